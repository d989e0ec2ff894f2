// libvdf_nova.so, part 2: the reference's Nova proof surface (src/nova/proof.rs:232-392) on top of the kernel ABI
// (include/vdf_hip.h): public parameters, circuits, prove_step / prove_recursively, verify.  Protocol
// "vdf-nova-ivc-v1", specified by oracle/nova.py; see include/vdf_nova.h.
#include "nova_internal.hpp"

using namespace vdfnova;

namespace vdfnova {
static thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

RelaxedInst to_relaxed(const Inst& in, const Field& own) {
  RelaxedInst r;
  r.comm_W = in.comm_W; r.comm_E = in.comm_E;
  fe_to_int(in.u, own, r.u);
  for (int k = 0; k < NUM_IO; ++k) fe_to_int(in.X[k], own, r.X[k]);
  return r;
}

std::unique_ptr<StepCircuit> make_primary_circuit(const vdf_pp* pp, const Circuit* c, bool device_rounds) {
  std::unique_ptr<InverseMinRootCircuit> m(new InverseMinRootCircuit());
  m->t = pp->t;
  m->bound = pp->circuit_kind == VDF_CIRCUIT_MINROOT_BOUND;
  m->device_rounds = device_rounds;
  m->blank = c == nullptr;                                              // blank circuit of the setup (:249-256)
  if (c) {
    m->result = MinRootState{c->result.x, c->result.y, c->result.i};
    m->input = MinRootState{c->input.x, c->input.y, c->input.i};
  } else {
    m->result = m->input = MinRootState{zero(), zero(), zero()};
  }
  return std::unique_ptr<StepCircuit>(m.release());
}
}  // namespace vdfnova

// ---- the step-circuit seam in the C ABI (include/vdf_nova.h) --------------------------------------------------------
struct vdf_cs { CS* cs; std::vector<Num> pool; bool bad = false; };
namespace {
const Num* cs_get(const vdf_cs* c, vdf_num h) { return h < c->pool.size() ? &c->pool[h] : nullptr; }
vdf_num cs_put(vdf_cs* c, Num n) { c->pool.push_back(std::move(n)); return (vdf_num)(c->pool.size() - 1); }
struct CustomStepCircuit : StepCircuit {
  vdf_step_circuit c;
  mutable int rc = 0;
  size_t arity() const override { return c.arity; }
  std::vector<Num> synthesize(CS& cs, const std::vector<Num>& z) const override {
    vdf_cs h{&cs, z, false};
    std::vector<vdf_num> zin(c.arity), zout(c.arity, 0);
    for (size_t k = 0; k < c.arity; ++k) zin[k] = (vdf_num)k;
    rc = c.synthesize(c.self, &h, zin.data(), zout.data());
    std::vector<Num> out(c.arity);
    for (size_t k = 0; k < c.arity; ++k) {
      const Num* n = cs_get(&h, zout[k]);
      if (!n || h.bad) { rc = rc ? rc : VDF_ERR_BAD_ARG; out[k] = cs.zero_num(); } else out[k] = *n;
    }
    return out;
  }
  void output(const Fe*, Fe*) const override {}        // z_{i+1} is the value of z_out after a witness synthesis
};
}  // namespace
namespace vdfnova {
std::unique_ptr<StepCircuit> make_custom_circuit(const vdf_step_circuit* c) {
  std::unique_ptr<CustomStepCircuit> m(new CustomStepCircuit());
  m->c = *c;
  return std::unique_ptr<StepCircuit>(m.release());
}
}  // namespace vdfnova

extern "C" {
int vdf_cs_is_witness(const vdf_cs* c) { return c && !c->cs->shape ? 1 : 0; }
vdf_num vdf_cs_const(vdf_cs* c, const vdf_fe* k) { Fe v; memcpy(&v, k, 32); return cs_put(c, c->cs->constant(v)); }
#define CS_BIN(name, op)                                                                      \
  vdf_num name(vdf_cs* c, vdf_num a, vdf_num b) {                                             \
    const Num *x = cs_get(c, a), *y = cs_get(c, b);                                           \
    if (!x || !y) { c->bad = true; return 0; }                                                \
    return cs_put(c, c->cs->op(*x, *y));                                                      \
  }
CS_BIN(vdf_cs_add, add)
CS_BIN(vdf_cs_sub, sub)
CS_BIN(vdf_cs_mul, mul)
#undef CS_BIN
vdf_num vdf_cs_scale(vdf_cs* c, vdf_num a, const vdf_fe* k) {
  const Num* x = cs_get(c, a);
  if (!x) { c->bad = true; return 0; }
  Fe v; memcpy(&v, k, 32);
  return cs_put(c, c->cs->scale(*x, v));
}
vdf_num vdf_cs_alloc(vdf_cs* c, const vdf_fe* value) {
  Fe v = zero();
  if (value && !c->cs->shape) memcpy(&v, value, 32);
  return cs_put(c, c->cs->alloc(v));
}
int vdf_cs_enforce(vdf_cs* c, vdf_num a, vdf_num b, vdf_num cc) {
  const Num *x = cs_get(c, a), *y = cs_get(c, b), *z = cs_get(c, cc);
  if (!x || !y || !z) { c->bad = true; return VDF_ERR_BAD_ARG; }
  c->cs->enforce(*x, *y, *z);
  return VDF_OK;
}
int vdf_cs_value(const vdf_cs* c, vdf_num a, vdf_fe* out) {
  const Num* x = cs_get(c, a);
  if (!x || !out) return VDF_ERR_BAD_ARG;
  memcpy(out, &x->v, 32);
  return VDF_OK;
}
}  // extern "C"

namespace {

AugInputs blank_inputs(size_t arity) {
  AugInputs in;
  in.params = in.i = zero();
  in.z0.assign(arity, zero());
  in.zi.assign(arity, zero());
  memset(&in.U, 0, sizeof(in.U));
  memset(&in.u_W, 0, sizeof(in.u_W));
  memset(in.u_X, 0, sizeof(in.u_X));
  memset(&in.T, 0, sizeof(in.T));
  return in;
}

struct HostShape { Coo m[3]; size_t num_cons = 0, num_vars = 0, step_begin = 0, step_end = 0; };

// both augmented circuits in shape mode (PublicParams::setup, src/nova/proof.rs:236)
int build_shapes(uint64_t t, int circuit_kind, HostShape out[2], const vdf_step_circuit* custom = nullptr, const RoInstance* ro = nullptr) {
  vdf_pp tmp;
  tmp.t = t;
  tmp.circuit_kind = circuit_kind;
  for (int side = 0; side < 2; ++side) {
    CS cs(side_field(side), true, ro);
    std::unique_ptr<StepCircuit> step;
    if (side == PRIMARY) step = custom ? make_custom_circuit(custom) : make_primary_circuit(&tmp, nullptr, false);
    else step.reset(new TrivialTestCircuit());
    // the step circuit's variables are one contiguous run: find it by synthesising the wrapper once around an empty step
    AugInputs blank = blank_inputs(step->arity());
    blank.ro = ro;
    synthesize_augmented(cs, side, blank, *step);
    if (side == PRIMARY && custom && static_cast<const CustomStepCircuit*>(step.get())->rc != 0)
      return fail(VDF_ERR_BAD_ARG, "the step circuit's synthesize failed while its shape was recorded");
    cs.finish(out[side].m);
    out[side].num_cons = cs.rows;
    out[side].num_vars = cs.W.size();
    out[side].step_begin = cs.step_begin;
    out[side].step_end = cs.step_end;
  }
  return VDF_OK;
}

// oracle/nova.py digest_shapes
void digest_shapes(uint64_t t, int gens_family, const HostShape sh[2], uint8_t out[32], const RoInstance* ro = nullptr) {
  Shake256 h;
  h.absorb("vdf-nova-ivc-v1", 15);
  if (!ro) ro = ro_default();
  h.absorb(ro->label.data(), ro->label.size());      // the random oracle's parameter block ("vdf-poseidon2-v1" for the default)
  const uint64_t hdr[3] = {t, GENS_SEED, (uint64_t)gens_family};
  h.absorb(hdr, sizeof(hdr));
  for (int side = 0; side < 2; ++side) {
    const Field& F = field(side_field(side));
    const uint64_t sz[3] = {(uint64_t)sh[side].num_cons, (uint64_t)sh[side].num_vars, (uint64_t)NUM_IO};
    h.absorb(sz, sizeof(sz));
    for (int k = 0; k < 3; ++k) {
      const Coo& m = sh[side].m[k];
      const uint64_t n = m.rows.size();
      h.absorb(&n, 8);
      std::vector<uint8_t> buf(n * 40);
      for (size_t e = 0; e < n; ++e) {
        memcpy(&buf[e * 40], &m.rows[e], 4);
        memcpy(&buf[e * 40 + 4], &m.cols[e], 4);
        const Fe c = from_mont(m.vals[e], F);
        memcpy(&buf[e * 40 + 8], c.l, 32);
      }
      h.absorb(buf.data(), buf.size());
    }
  }
  h.squeeze(out, 32);
  out[31] &= 0x03;                                 // 250 bits
}

// the running instance a circuit hands back: nine native values -> (commitments, u, X) in the instance's own field
Inst inst_from_elements(const Fe e[9], const Field& circuit_field, const Field& own) {
  Inst in;
  in.comm_W = Aff{e[0], e[1]};
  in.comm_E = Aff{e[2], e[3]};
  uint64_t v[4];
  fe_to_int(e[4], circuit_field, v);
  in.u = int_to_fe(v, own);
  for (int k = 0; k < 2; ++k) {
    uint64_t lo[4], hi[4], x[4] = {0, 0, 0, 0};
    fe_to_int(e[5 + 2 * k], circuit_field, lo);
    fe_to_int(e[6 + 2 * k], circuit_field, hi);
    // x = lo + hi * 2^126 (< 2^255)
    x[0] = lo[0]; x[1] = lo[1] | (hi[0] << 62); x[2] = (hi[0] >> 2) | (hi[1] << 62); x[3] = (hi[1] >> 2) | (hi[2] << 62);
    in.X[k] = int_to_fe(x, own);
  }
  return in;
}

}  // namespace

namespace vdfnova {

int alloc_proof_buffers(vdf_proof* p) {
  vdf_pp* pp = p->pp;
  vdf_ctx* ctx = pp->ctx;
  for (int s = 0; s < 2; ++s) {
    const Side& sd = pp->s[s];
    SideState& st = p->r[s];
    HIPCALL(ctx, vdf_dev_alloc(ctx, sd.ncols * 32, &st.d_z));
    HIPCALL(ctx, vdf_dev_alloc(ctx, sd.num_cons * 32, &st.d_E));
    HIPCALL(ctx, vdf_dev_alloc(ctx, sd.num_cons * 32, &st.d_T));
    for (int k = 0; k < 3; ++k) {
      HIPCALL(ctx, vdf_dev_alloc(ctx, sd.num_cons * 32, &st.d_abc[k]));
      HIPCALL(ctx, vdf_dev_alloc(ctx, sd.num_cons * 32, &st.d_abc2[k]));
    }
    HIPCALL(ctx, vdf_dev_memset(ctx, st.d_z, 0, sd.ncols * 32));
    HIPCALL(ctx, vdf_dev_memset(ctx, st.d_E, 0, sd.num_cons * 32));
    for (int k = 0; k < 3; ++k) HIPCALL(ctx, vdf_dev_memset(ctx, st.d_abc[k], 0, sd.num_cons * 32));
    HIPCALL(ctx, vdf_host_alloc(ctx, (sd.ncols - (s == PRIMARY ? pp->seg_len : 0)) * 32, (void**)&p->h_stage[s]));
  }
  HIPCALL(ctx, vdf_dev_alloc(ctx, pp->s[SECONDARY].ncols * 32, &p->d_l2z));
  for (int k = 0; k < vdf_proof::RING; ++k) {
    HIPCALL(ctx, vdf_dev_alloc(ctx, pp->s[PRIMARY].ncols * 32, &p->d_z2s[k]));
    // a fresh instance has u = 1: in place from the start, because the early rows of a cross term read it before the
    // host's share of that witness is uploaded
    const Fe u_one = one(*pp->s[PRIMARY].F);
    HIPCALL(ctx, vdf_dev_memcpy(ctx, (char*)p->d_z2s[k] + pp->s[PRIMARY].num_vars * 32, &u_one, 32));
  }
  for (int k = 0; k < vdf_proof::DEPTH; ++k) {
    if (pp->seg_gens) HIPCALL(ctx, vdf_dev_alloc(ctx, (3 * pp->t + 4) * 32, &p->d_packed[k]));
    const int dev = vdf_ctx_device(ctx);
    if ((k == 0 ? vdf_ctx_create_pooled_near(ctx, VDF_QUEUE_SIDE, &p->ctx2[k]) : vdf_ctx_create_pooled(&dev, 1, VDF_QUEUE_SIDE, &p->ctx2[k])) != VDF_OK)
      return fail(VDF_ERR_DEVICE, std::string("lookahead context: ") + vdf_last_error(nullptr));
    HIPCALL(p->ctx2[k], vdf_ctx_set_async(p->ctx2[k], 1));
    // the lookahead's commitment is needed a whole step later: its sort and bucket reduction yield to the early rows' (the
    // step's longest dependent path, priority 3) and to the chain's direct sums (2).  vdf_nova_tuning.lookahead_priority
    HIPCALL(p->ctx2[k], vdf_ctx_set_light_priority(p->ctx2[k], pp->tune.lookahead_priority));
    // both side queues' bucket accumulations share the device with other queues' kernels: a grid sized for all three resident
    // workgroups per CU (a grid sized for two is packed three-and-one by the dispatcher; r4: 0.917 -> 0.867 ms per step)
    HIPCALL(p->ctx2[k], vdf_ctx_set_accumulate_fill(p->ctx2[k], pp->tune.side_accumulate_fill));
  }
  {
    if (vdf_ctx_create_pooled_near(ctx, VDF_QUEUE_CRITICAL, &p->ctx3) != VDF_OK) return fail(VDF_ERR_DEVICE, std::string("early-rows context: ") + vdf_last_error(nullptr));
    HIPCALL(p->ctx3, vdf_ctx_set_async(p->ctx3, 1));
    HIPCALL(p->ctx3, vdf_ctx_set_accumulate_fill(p->ctx3, pp->tune.side_accumulate_fill));
  }
  HIPCALL(ctx, vdf_host_alloc(ctx, (vdf_proof::RING + 8) * sizeof(vdf_jac), (void**)&p->h_pts));
  // two pinned slots: [0, arity) this step's z_in when its early rows start with the step, [arity, 2 arity) the NEXT step's, written
  // half a step early -- one buffer had two writers whose async copies were not ordered against each other
  HIPCALL(ctx, vdf_host_alloc(ctx, 2 * pp->arity * 32, (void**)&p->h_zin));
  memset(&p->last, 0, sizeof(p->last));
  return VDF_OK;
}

// commitment of the last secondary witness (its MSM normally rides in the next step's batch)
int finalize_l2(const vdf_proof* cp) {
  vdf_proof* p = const_cast<vdf_proof*>(cp);
  if (p->poisoned) return fail(VDF_ERR_DEVICE, "this proof's running instance is half folded (an earlier prove_step failed on the device)");
  const Side& sd = p->pp->s[SECONDARY];
  vdf_ctx* ctx = sd.ctx;
  if (p->nifs2 == vdf_proof::NIFS2_INFLIGHT) {        // the step launched it on its way out: collect
    vdf_jac* hb = &p->h_pts[vdf_proof::RING];
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    if (!p->l2_committed) { p->l2.comm_W = jac_to_aff(hb[0], *sd.Fb); p->l2_committed = true; }
    p->nifs2_T = jac_to_aff(hb[1], *sd.Fb);
    p->nifs2 = vdf_proof::NIFS2_DONE;
  }
  if (p->l2_committed) return VDF_OK;
  vdf_jac* slot = &p->h_pts[vdf_proof::RING];
  HIPCALL(ctx, vdf_msm(ctx, sd.gens, 0, (const vdf_fe*)p->d_l2z, sd.num_vars, 1, slot));
  HIPCALL(ctx, vdf_ctx_sync(ctx));
  p->l2.comm_W = jac_to_aff(*slot, *sd.Fb);
  p->l2_committed = true;
  return VDF_OK;
}

int check_sat(const Side& sd, const Inst& in, const void* d_z, const void* d_E, void* d_abc[3], void* d_T, bool* ok) {
  vdf_ctx* ctx = sd.ctx;
  *ok = false;
  vdf_jac j1, j2;
  HIPCALL(ctx, vdf_msm(ctx, sd.gens, 0, (const vdf_fe*)d_z, sd.num_vars, 1, &j1));
  const Aff a1 = jac_to_aff(j1, *sd.Fb);
  if (memcmp(&a1, &in.comm_W, 64)) return VDF_OK;
  if (d_E) {
    HIPCALL(ctx, vdf_msm(ctx, sd.gens, 0, (const vdf_fe*)d_E, sd.num_cons, 1, &j2));
    const Aff a2 = jac_to_aff(j2, *sd.Fb);
    if (memcmp(&a2, &in.comm_E, 64)) return VDF_OK;
  } else if (!in.comm_E.is_id()) return VDF_OK;
  // z = (W, u, X) must carry the instance's u and X
  std::vector<Fe> tail(1 + NUM_IO);
  HIPCALL(ctx, vdf_dev_memcpy(ctx, tail.data(), (const char*)d_z + sd.num_vars * 32, tail.size() * 32));
  if (tail[0] != in.u || memcmp(&tail[1], in.X, 32 * NUM_IO)) return VDF_OK;
  HIPCALL(ctx, vdf_spmv3(ctx, sd.shape, (const vdf_fe*)d_z, (vdf_fe*)d_abc[0], (vdf_fe*)d_abc[1], (vdf_fe*)d_abc[2]));
  // residual Az*Bz - u*Cz - E through the cross-term kernel with Az2 = Bz1 = 0 and "Cz1" = E
  HIPCALL(ctx, vdf_cross_term(ctx, sd.field, (const vdf_fe*)d_abc[0], (const vdf_fe*)sd.d_zero, (const vdf_fe*)(d_E ? d_E : sd.d_zero),
                              (const vdf_fe*)sd.d_zero, (const vdf_fe*)d_abc[1], (const vdf_fe*)d_abc[2], (const vdf_fe*)&in.u,
                              sd.num_cons, (vdf_fe*)d_T));
  int zero = 0;
  HIPCALL(ctx, vdf_vec_is_zero(ctx, (const vdf_fe*)d_T, sd.num_cons, &zero));
  *ok = zero != 0;
  return VDF_OK;
}

}  // namespace vdfnova

extern "C" {

const char* vdf_nova_last_error(void) { return vdfnova::g_err.c_str(); }


// ---- the random oracle's parameter block ------------------------------------------------------------------------
// null -> the default; a block this build does not support -> nullptr with *bad set
static const RoInstance* ro_from_abi(const vdf_nova_ro_params* r, bool* bad) {
  *bad = false;
  if (!r) return ro_default();
  if (r->struct_size != sizeof(vdf_nova_ro_params)) { *bad = true; return nullptr; }
  RoSpec sp;
  sp.family = r->family; sp.width = r->width; sp.full_rounds = r->full_rounds; sp.partial_rounds = r->partial_rounds;
  sp.alpha = r->alpha; sp.challenge_bits = r->challenge_bits; sp.hash_bits = r->hash_bits;
  const RoInstance* ro = ro_instance(sp);
  if (!ro) *bad = true;
  return ro;
}
static void ro_to_abi(const RoInstance* ro, vdf_nova_ro_params* out) {
  const RoSpec& sp = (ro ? ro : ro_default())->spec;
  out->struct_size = (uint32_t)sizeof(vdf_nova_ro_params);
  out->family = sp.family; out->width = sp.width; out->full_rounds = sp.full_rounds; out->partial_rounds = sp.partial_rounds;
  out->alpha = sp.alpha; out->challenge_bits = sp.challenge_bits; out->hash_bits = sp.hash_bits;
}
int vdf_nova_ro_preset(int which, vdf_nova_ro_params* out) {
  if (!out || (which != 0 && which != 1)) return fail(VDF_ERR_BAD_ARG, "bad argument");
  RoInstance tmp;
  if (which == 1) { tmp.spec.family = VDF_RO_POSEIDON; tmp.spec.width = 25; tmp.spec.full_rounds = 8; tmp.spec.partial_rounds = 57; }
  ro_to_abi(&tmp, out);
  return VDF_OK;
}
int vdf_nova_pp_ro(const vdf_pp* pp, vdf_nova_ro_params* out) {
  if (!pp || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
  ro_to_abi(pp->ro, out);
  return VDF_OK;
}

// ---- host-only entry points ------------------------------------------------------------------------------------
int vdf_nova_ro_hash_ro(const vdf_nova_ro_params* rop, int f, uint64_t tag, const vdf_fe* xs, size_t n, vdf_fe* out) {
  return nova_guard([&]() -> int {
    bool bad;
    const RoInstance* ro = ro_from_abi(rop, &bad);
    if (bad) return fail(VDF_ERR_BAD_ARG, "unsupported RO parameter block");
    if (!valid_field(f) || (!xs && n) || !out) return fail(VDF_ERR_BAD_ARG, "bad argument");
    const Fe r = ro_hash(f, tag, (const Fe*)xs, n, ro);
    memcpy(out, &r, 32);
    return VDF_OK;
  });
}
int vdf_nova_ro_hash(int f, uint64_t tag, const vdf_fe* xs, size_t n, vdf_fe* out) { return vdf_nova_ro_hash_ro(nullptr, f, tag, xs, n, out); }

int vdf_nova_shape_digest(uint64_t t, int circuit_kind, int gens_family, uint8_t out[32], uint64_t sizes[2][3]) {
  return vdf_nova_shape_digest_ro(nullptr, t, circuit_kind, gens_family, out, sizes);
}
int vdf_nova_shape_digest_ro(const vdf_nova_ro_params* rop, uint64_t t, int circuit_kind, int gens_family, uint8_t out[32], uint64_t sizes[2][3]) {
  return nova_guard([&]() -> int {
    bool bad;
    const RoInstance* ro = ro_from_abi(rop, &bad);
    if (bad) return fail(VDF_ERR_BAD_ARG, "unsupported RO parameter block");
    if (t == 0 || t > (1ull << 24) || !out) return fail(VDF_ERR_BAD_ARG, "bad argument");
    HostShape sh[2];
    build_shapes(t, circuit_kind, sh, nullptr, ro);
    digest_shapes(t, gens_family, sh, out, ro);
    if (sizes)
      for (int s = 0; s < 2; ++s) {
        sizes[s][0] = sh[s].num_cons; sizes[s][1] = sh[s].num_vars;
        sizes[s][2] = sh[s].m[0].rows.size() + sh[s].m[1].rows.size() + sh[s].m[2].rows.size();
      }
    return VDF_OK;
  });
}

// The R1CS shape public_params would make, as COO triples (row-major order of the constraints' creation, values in Montgomery
// form of the side's field): two calls, first with null arrays for the counts.
int vdf_nova_shape_export(uint64_t t, int circuit_kind, int side, uint64_t nnz[3], uint32_t* const rows[3], uint32_t* const cols[3],
                          vdf_fe* const vals[3]) {
  return nova_guard([&]() -> int {
    if (t == 0 || t > (1ull << 24) || !nnz || (side != PRIMARY && side != SECONDARY)) return fail(VDF_ERR_BAD_ARG, "bad argument");
    if (circuit_kind != VDF_CIRCUIT_MINROOT_BOUND && circuit_kind != VDF_CIRCUIT_MINROOT_REFERENCE) return fail(VDF_ERR_BAD_ARG, "unknown step circuit");
    HostShape sh[2];
    build_shapes(t, circuit_kind, sh);
    const HostShape& h = sh[side];
    const bool fill = rows && cols && vals;
    for (int k = 0; k < 3; ++k) {
      const size_t z = h.m[k].rows.size();
      if (fill) {
        if (nnz[k] < z || !rows[k] || !cols[k] || !vals[k]) return fail(VDF_ERR_BAD_LENGTH, "triple arrays too short");
        memcpy(rows[k], h.m[k].rows.data(), z * 4);
        memcpy(cols[k], h.m[k].cols.data(), z * 4);
        memcpy(vals[k], h.m[k].vals.data(), z * 32);
      }
      nnz[k] = z;
    }
    return VDF_OK;
  });
}

int vdf_nova_shape_digest_custom(const vdf_step_circuit* primary, int gens_family, uint8_t out[32], uint64_t sizes[2][3]) {
  return nova_guard([&]() -> int {
    if (!primary || !primary->synthesize || primary->arity == 0 || primary->arity > 64 || !out) return fail(VDF_ERR_BAD_ARG, "bad argument");
    HostShape sh[2];
    { int rc = build_shapes(0, VDF_CIRCUIT_CUSTOM, sh, primary); if (rc != VDF_OK) return rc; }
    digest_shapes(0, gens_family, sh, out);
    if (sizes)
      for (int s = 0; s < 2; ++s) {
        sizes[s][0] = sh[s].num_cons; sizes[s][1] = sh[s].num_vars;
        sizes[s][2] = sh[s].m[0].rows.size() + sh[s].m[1].rows.size() + sh[s].m[2].rows.size();
      }
    return VDF_OK;
  });
}

static AugInputs aug_from_abi(int side, const vdf_nova_aug_inputs* a) {
  const Field& own = field(side_field(1 - side));                       // the folded side's scalar field
  const size_t arity = side == PRIMARY ? 3 : 1;
  AugInputs in;
  memcpy(&in.params, &a->params, 32);
  memcpy(&in.i, &a->i, 32);
  in.z0.resize(arity); in.zi.resize(arity);
  memcpy(in.z0.data(), a->z0, 32 * arity);
  memcpy(in.zi.data(), a->zi, 32 * arity);
  Inst U;
  memcpy(&U.comm_W, &a->U_comm_W, 64); memcpy(&U.comm_E, &a->U_comm_E, 64);
  memcpy(&U.u, &a->U_u, 32); memcpy(U.X, a->U_X, 64);
  in.U = to_relaxed(U, own);
  memcpy(&in.u_W, &a->u_comm_W, 64);
  for (int k = 0; k < 2; ++k) { Fe x; memcpy(&x, &a->u_X[k], 32); fe_to_int(x, own, in.u_X[k]); }
  memcpy(&in.T, &a->T, 64);
  return in;
}

int vdf_nova_aug_synthesize(int side, uint64_t t, int circuit_kind, const vdf_nova_aug_inputs* a, const vdf_state* result,
                            const vdf_state* input, vdf_fe* W, size_t w_cap, size_t* num_vars, size_t* num_cons, vdf_fe X[2],
                            vdf_fe z_next[3]) {
  return vdf_nova_aug_synthesize_ro(nullptr, side, t, circuit_kind, a, result, input, W, w_cap, num_vars, num_cons, X, z_next);
}
int vdf_nova_aug_synthesize_ro(const vdf_nova_ro_params* rop, int side, uint64_t t, int circuit_kind, const vdf_nova_aug_inputs* a,
                               const vdf_state* result, const vdf_state* input, vdf_fe* W, size_t w_cap, size_t* num_vars, size_t* num_cons,
                               vdf_fe X[2], vdf_fe z_next[3]) {
  return nova_guard([&]() -> int {
    bool bad;
    const RoInstance* ro = ro_from_abi(rop, &bad);
    if (bad) return fail(VDF_ERR_BAD_ARG, "unsupported RO parameter block");
    if ((side != PRIMARY && side != SECONDARY) || !a || t == 0) return fail(VDF_ERR_BAD_ARG, "bad argument");
    vdf_pp tmp;
    tmp.t = t;
    tmp.circuit_kind = circuit_kind;
    std::unique_ptr<StepCircuit> step;
    Circuit c;
    if (side == PRIMARY) {
      if (!result || !input) return fail(VDF_ERR_BAD_ARG, "the primary circuit needs the step's states");
      c.result = load_state(result); c.input = load_state(input); c.t = t;
      step = make_primary_circuit(&tmp, &c, false);
    } else step.reset(new TrivialTestCircuit());
    CS cs(side_field(side), false, ro);
    AugInputs ain = aug_from_abi(side, a);
    ain.ro = ro;
    const std::vector<Fe> zn = synthesize_augmented(cs, side, ain, *step);
    if (num_vars) *num_vars = cs.W.size();
    if (num_cons) *num_cons = cs.rows;
    if (W) {
      if (w_cap < cs.W.size()) return fail(VDF_ERR_BAD_LENGTH, "W buffer too small");
      memcpy(W, cs.W.data(), cs.W.size() * 32);
    }
    if (X) memcpy(X, cs.X.data(), 64);
    if (z_next) memcpy(z_next, zn.data(), zn.size() * 32);
    return VDF_OK;
  });
}

int vdf_nova_synthesis_stats(uint64_t* queued, uint64_t* misses) {
  if (!queued || !misses) return fail(VDF_ERR_BAD_ARG, "null argument");
  last_synthesis_stats(queued, misses);
  return VDF_OK;
}

// ---- tuning --------------------------------------------------------------------------------------------
}  // extern "C"
namespace vdfnova {
bool tuning_valid(const vdf_nova_tuning& t) {
  auto in = [](int v, int lo, int hi) { return v >= lo && v <= hi; };
  return (t.flags & ~(uint32_t)(VDF_PP_NO_DIGIT_TABLES | VDF_PP_NO_EARLY_ROWS)) == 0 &&
         (t.digit_window == -1 || t.digit_window == 0 || in(t.digit_window, 6, 12)) && in(t.early_rows, 0, 2) && in(t.stencil, 0, 1) &&
         in(t.small_window, 6, 16) && in(t.big_window, 12, 20) && in(t.packed_commit, 0, 1) && in(t.lookahead_early, 0, 1) &&
         in(t.gate_accumulate, 0, 1) && in(t.fold_on_rows, 0, 1) && in(t.nifs_ahead, 0, 1) && in(t.early_row_parts, 1, 3) &&
         in(t.lookahead_priority, 0, 3) && in(t.side_accumulate_fill, 1, 3) && in(t.verbose, 0, 1) && in(t.compress_queues, 0, 1) && in(t.rows_at_challenge, 0, 1) &&
         in(t.fold_fused, 0, 1);
}
const vdf_nova_tuning& default_tuning() {
  static const vdf_nova_tuning d = [] {
    vdf_nova_tuning t{};
    t.struct_size = (uint32_t)sizeof(vdf_nova_tuning);
    t.flags = 0; t.digit_budget_bytes = (uint64_t)20 << 30; t.digit_window = 0; t.early_rows = 2; t.stencil = 1; t.small_window = 15;
    t.big_window = 16; t.packed_commit = 1; t.lookahead_early = 1; t.gate_accumulate = 1; t.fold_on_rows = 1; t.nifs_ahead = 1;
    t.early_row_parts = 1; t.lookahead_priority = 1; t.side_accumulate_fill = 3; t.verbose = 0; t.compress_queues = 1; t.rows_at_challenge = 1;
    t.fold_fused = 0;      // measured (profiles/r05_ab_fold_fused.txt): the fused fold shortens the rows' path by ~45 us and the step gets no faster
    // the environment overrides of earlier rounds, read once: the only place the prover looks at the environment for tuning
    const struct { const char* name; int32_t* field; } vars[] = {
        {"VDF_NOVA_DIGIT_WINDOW", &t.digit_window}, {"VDF_NOVA_T_AHEAD", &t.early_rows}, {"VDF_NOVA_STENCIL", &t.stencil},
        {"VDF_NOVA_SMALL_WINDOW", &t.small_window}, {"VDF_NOVA_BIG_WINDOW", &t.big_window}, {"VDF_NOVA_PACKED_COMMIT", &t.packed_commit},
        {"VDF_NOVA_LOOKAHEAD_EARLY", &t.lookahead_early}, {"VDF_NOVA_GATE", &t.gate_accumulate}, {"VDF_NOVA_FOLD_ON_ROWS", &t.fold_on_rows},
        {"VDF_NOVA_NIFS_AHEAD", &t.nifs_ahead}, {"VDF_NOVA_T_PARTS", &t.early_row_parts}, {"VDF_NOVA_LOOKAHEAD_PRIO", &t.lookahead_priority},
        {"VDF_NOVA_SIDE_ACC_WG", &t.side_accumulate_fill}, {"VDF_NOVA_VERBOSE", &t.verbose}, {"VDF_NOVA_COMPRESS_QUEUES", &t.compress_queues}, {"VDF_NOVA_ROWS_AT_CHALLENGE", &t.rows_at_challenge},
        {"VDF_NOVA_FOLD_FUSED", &t.fold_fused}};
    for (const auto& v : vars) {
      const char* e = env_override(v.name);
      if (!e || !*e) continue;
      const int32_t old = *v.field;
      *v.field = (int32_t)atol(e);
      if (v.field == &t.digit_window && *v.field == 0) *v.field = -1;            // VDF_NOVA_DIGIT_WINDOW=0 has always meant "none"
      if (!tuning_valid(t)) *v.field = old;                                     // an out-of-range override is ignored
    }
    if (const char* e = env_override("VDF_NOVA_DIGIT_BUDGET_GIB")) { const long g = atol(e); if (g >= 0 && g <= 1024) t.digit_budget_bytes = (uint64_t)g << 30; }
    return t;
  }();
  return d;
}
}  // namespace vdfnova
extern "C" {
void vdf_nova_tuning_default(vdf_nova_tuning* out) { if (out) *out = default_tuning(); }
int vdf_nova_pp_tuning(const vdf_pp* pp, vdf_nova_tuning* out) {
  if (!pp || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
  *out = pp->tune;
  return VDF_OK;
}
int vdf_nova_pp_setup_ms(const vdf_pp* pp, double ms[7]) {
  if (!pp || !ms) return fail(VDF_ERR_BAD_ARG, "null argument");
  memcpy(ms, pp->setup_ms, sizeof(pp->setup_ms));
  return VDF_OK;
}

// ---- public parameters -------------------------------------------------------------------------------
int vdf_nova_public_params(vdf_ctx* ctx, uint64_t t, vdf_pp** out) {
  return nova_guard([&]() -> int {
    // the reference's own step circuit (src/nova/proof.rs:155-230), as public_params(num_iters_per_step) builds it (:232-237)
    return vdf_nova_public_params_ex(ctx, t, VDF_CIRCUIT_MINROOT_REFERENCE, VDF_GENS_TRY_AND_INCREMENT, out);
  });
}

static int public_params_impl(vdf_ctx* ctx, uint64_t t, int circuit_kind, const vdf_step_circuit* custom, int gens_family,
                              const vdf_nova_tuning& tune, vdf_pp** out, const RoInstance* ro = nullptr);

int vdf_nova_public_params_tuned(vdf_ctx* ctx, uint64_t t, int circuit_kind, int gens_family, const vdf_nova_tuning* tuning, vdf_pp** out) {
  return nova_guard([&]() -> int {
    if (!ctx || !out || t == 0 || t > (1ull << 24)) return fail(VDF_ERR_BAD_ARG, "bad argument");
    if (circuit_kind != VDF_CIRCUIT_MINROOT_BOUND && circuit_kind != VDF_CIRCUIT_MINROOT_REFERENCE)
      return fail(VDF_ERR_BAD_ARG, "unknown step circuit");
    if (tuning && (tuning->struct_size != sizeof(vdf_nova_tuning) || !tuning_valid(*tuning))) return fail(VDF_ERR_BAD_ARG, "tuning: a field is out of range");
    return public_params_impl(ctx, t, circuit_kind, nullptr, gens_family, tuning ? *tuning : default_tuning(), out);
  });
}
int vdf_nova_public_params_ro(vdf_ctx* ctx, uint64_t t, int circuit_kind, int gens_family, const vdf_nova_ro_params* rop,
                              const vdf_nova_tuning* tuning, vdf_pp** out) {
  return nova_guard([&]() -> int {
    bool bad;
    const RoInstance* ro = ro_from_abi(rop, &bad);
    if (bad) return fail(VDF_ERR_BAD_ARG, "unsupported RO parameter block (vdf_nova.h: alpha 5, 128 / 250 bits, family 1 widths 2..25)");
    if (!ctx || !out || t == 0 || t > (1ull << 24)) return fail(VDF_ERR_BAD_ARG, "bad argument");
    if (circuit_kind != VDF_CIRCUIT_MINROOT_BOUND && circuit_kind != VDF_CIRCUIT_MINROOT_REFERENCE)
      return fail(VDF_ERR_BAD_ARG, "unknown step circuit");
    if (tuning && (tuning->struct_size != sizeof(vdf_nova_tuning) || !tuning_valid(*tuning))) return fail(VDF_ERR_BAD_ARG, "tuning: a field is out of range");
    return public_params_impl(ctx, t, circuit_kind, nullptr, gens_family, tuning ? *tuning : default_tuning(), out, ro);
  });
}
int vdf_nova_public_params_flags(vdf_ctx* ctx, uint64_t t, int circuit_kind, int gens_family, uint32_t flags, vdf_pp** out) {
  if (flags & ~(uint32_t)(VDF_PP_NO_DIGIT_TABLES | VDF_PP_NO_EARLY_ROWS)) return fail(VDF_ERR_BAD_ARG, "unknown flag");
  vdf_nova_tuning tn = default_tuning();
  tn.flags |= flags;
  return vdf_nova_public_params_tuned(ctx, t, circuit_kind, gens_family, &tn, out);
}
int vdf_nova_public_params_ex(vdf_ctx* ctx, uint64_t t, int circuit_kind, int gens_family, vdf_pp** out) {
  return vdf_nova_public_params_flags(ctx, t, circuit_kind, gens_family, 0, out);
}

int vdf_nova_public_params_custom(vdf_ctx* ctx, const vdf_step_circuit* primary, int gens_family, vdf_pp** out) {
  return nova_guard([&]() -> int {
    if (!ctx || !out || !primary || !primary->synthesize || primary->arity == 0 || primary->arity > 64)
      return fail(VDF_ERR_BAD_ARG, "bad argument");
    return public_params_impl(ctx, 0, VDF_CIRCUIT_CUSTOM, primary, gens_family, default_tuning(), out);
  });
}

// The reference's circuit allocates new_x in every round (src/nova/proof.rs:167-173) although it is an affine image of
// other witness values: new_x_j = y_j - (i_in - (j + 1)), y_j = new_y_(j-1).  Its share of a commitment therefore folds
// into the generators of the new_y's and two points that depend on the generators only (include/vdf_hip.h
// vdf_minroot_step_segment_packed): the MinRoot rounds of the reference's witness are committed by an MSM of 3t + 4 terms
// over DERIVED generators instead of 4t + 1 -- the same group element, a quarter fewer bucket additions.
static int make_packed_generators(vdf_pp* pp) {
  const Side& sd = pp->s[PRIMARY];
  const Field& Fb = *sd.Fb;
  const size_t t = pp->t, n = 4 * t + 1;
  std::vector<Aff> G(n);
  HIPCALL(pp->ctx, vdf_bases_download(pp->ctx, sd.gens, pp->seg_begin, n, (vdf_affine*)G.data()));
  std::vector<Aff> D(3 * t + 4);
  std::vector<Pt> sums(t + 2);                    // G[4j+3] + G[4j+4] for j < t - 1, then S1 (negated below) and S2
  Pt s1 = pt_identity(), s2 = pt_identity();
  for (size_t j = t; j-- > 0;) {                  // suffix sums: S2 = sum_j (j + 1) G[4j]
    s1 = pt_add(s1, pt_from_aff(G[4 * j], Fb), Fb);
    s2 = pt_add(s2, s1, Fb);
  }
  for (size_t j = 0; j + 1 < t; ++j) sums[j] = pt_add(pt_from_aff(G[4 * j + 3], Fb), pt_from_aff(G[4 * j + 4], Fb), Fb);
  sums[t - 1] = pt_from_aff(G[4 * t - 1], Fb);
  s1.y = neg(s1.y, Fb);
  sums[t] = s1; sums[t + 1] = s2;
  // XYZZ -> affine with two batched inversions
  std::vector<Fe> izz(t + 2), izzz(t + 2);
  for (size_t j = 0; j < t + 2; ++j) { izz[j] = sums[j].is_id() ? one(Fb) : sums[j].zz; izzz[j] = sums[j].is_id() ? one(Fb) : sums[j].zzz; }
  batch_inverse(izz.data(), izz.size(), Fb);
  batch_inverse(izzz.data(), izzz.size(), Fb);
  auto aff = [&](size_t j) {
    Aff a;
    if (sums[j].is_id()) { a.x = zero(); a.y = zero(); return a; }
    a.x = vdfhost::canon(vdfhost::mul(sums[j].x, izz[j], Fb), Fb); a.y = vdfhost::canon(vdfhost::mul(sums[j].y, izzz[j], Fb), Fb);
    return a;
  };
  for (size_t j = 0; j < t; ++j) { D[3 * j] = G[4 * j + 1]; D[3 * j + 1] = G[4 * j + 2]; D[3 * j + 2] = aff(j); }
  D[3 * t] = G[4 * t]; D[3 * t + 1] = G[0]; D[3 * t + 2] = aff(t); D[3 * t + 3] = aff(t + 1);
  HIPCALL(pp->ctx, vdf_bases_upload(pp->ctx, sd.curve, (const vdf_affine*)D.data(), D.size(), &pp->seg_gens));
  HIPCALL(pp->ctx, vdf_bases_precompute(pp->ctx, pp->seg_gens, vdf_bases_window(sd.gens), 1));
  return VDF_OK;
}

// the longest run of constraints that read nothing of a witness but [seg_begin - arity, seg_begin + seg_len) and the constant, none
// of them longer than 8 entries in any matrix (public_params_impl)
static void longest_early_run(const HostShape& h, size_t seg_begin, size_t seg_len, size_t arity, size_t* best_b, size_t* best_n) {
  const size_t sb = seg_begin - arity, se = seg_begin + seg_len;
  std::vector<uint8_t> early(h.num_cons, 1);
  for (int k = 0; k < 3; ++k) {
    std::vector<uint32_t> per_row(h.num_cons, 0);
    for (size_t e = 0; e < h.m[k].rows.size(); ++e) {
      const uint32_t r = h.m[k].rows[e], c = h.m[k].cols[e];
      if (!((c >= sb && c < se) || c == h.num_vars) || ++per_row[r] > 8) early[r] = 0;
    }
  }
  *best_b = 0; *best_n = 0;
  size_t run_b = 0;
  for (size_t r = 0; r <= h.num_cons; ++r)
    if (r == h.num_cons || !early[r]) { if (r - run_b > *best_n) { *best_b = run_b; *best_n = r - run_b; } run_b = r + 1; }
}

// Are the rows [row0, row0 + 3t + 1) of the primary shape EXACTLY the stencil vdf_nifs_cross_term_minroot computes
// (include/vdf_hip.h; src/nova/proof.rs:107-133, :219-227)?  Every triple of the three matrices in those rows is compared
// with the stencil's own list: a circuit that differs in one coefficient keeps the generic sparse kernel.
static bool minroot_stencil_matches(const HostShape& h, const Field& F, uint64_t t, int per, size_t S, size_t row0) {
  const size_t nrows = 3 * (size_t)t + 1, one_col = h.num_vars;
  if (S < 3 || row0 + nrows > h.num_cons || S + (size_t)per * t + 1 > h.num_vars) return false;
  typedef std::vector<std::pair<uint32_t, Fe>> Row;
  const Fe p1 = one(F), m1 = neg(one(F), F);
  for (int k = 0; k < 3; ++k) {
    std::vector<Row> got(nrows);
    const Coo& m = h.m[k];
    for (size_t e = 0; e < m.rows.size(); ++e)
      if (m.rows[e] >= row0 && m.rows[e] < row0 + nrows) got[m.rows[e] - row0].push_back({m.cols[e], vdfhost::canon(m.vals[e], F)});
    for (size_t i = 0; i < nrows; ++i) {
      Row want;
      if (i == nrows - 1) {
        if (k == 0) want = {{(uint32_t)(S + (size_t)per * t), p1}};
        else if (k == 1) want = {{(uint32_t)one_col, p1}};
        else want = {{(uint32_t)(S - 1), p1}, {(uint32_t)one_col, neg(from_u64(t, F), F)}};
      } else {
        const size_t j = i / 3, role = i - 3 * j, rd = S + (size_t)per * j, t1 = rd + (per - 3);
        const uint32_t y_j = (uint32_t)(j ? rd - 1 : S - 2);
        Row x;                                                       // the linear combination that is x_j
        if (per == 4 || j == 0) x = {{(uint32_t)(j ? rd - per : S - 3), p1}};
        else x = {{(uint32_t)(j > 1 ? rd - per - 1 : S - 2), p1}, {(uint32_t)(S - 1), m1}, {(uint32_t)one_col, from_u64(j, F)}};
        if (role == 0) want = k < 2 ? x : Row{{(uint32_t)t1, p1}};
        else if (role == 1) want = k < 2 ? Row{{(uint32_t)t1, p1}} : Row{{(uint32_t)(t1 + 1), p1}};
        else if (k == 0) want = {{(uint32_t)(t1 + 1), p1}};
        else if (k == 1) want = x;
        else want = {{(uint32_t)(t1 + 2), p1}, {y_j, p1}, {(uint32_t)(S - 1), m1}, {(uint32_t)one_col, from_u64(j + 1, F)}};
      }
      Row& g = got[i];
      if (g.size() != want.size()) return false;
      auto by_col = [](const std::pair<uint32_t, Fe>& a, const std::pair<uint32_t, Fe>& b) { return a.first < b.first; };
      std::sort(g.begin(), g.end(), by_col);
      std::sort(want.begin(), want.end(), by_col);
      for (size_t e = 0; e < g.size(); ++e)
        if (g[e].first != want[e].first || g[e].second != vdfhost::canon(want[e].second, F)) return false;
    }
  }
  return true;
}

static int public_params_impl(vdf_ctx* ctx, uint64_t t, int circuit_kind, const vdf_step_circuit* custom, int gens_family,
                              const vdf_nova_tuning& tune, vdf_pp** out, const RoInstance* ro) {
  if (gens_family != VDF_GENS_TRY_AND_INCREMENT && gens_family != VDF_GENS_KNOWN_DLOG && gens_family != VDF_GENS_LABEL_SHAKE)
    return fail(VDF_ERR_BAD_ARG, "unknown generator family");
  *out = nullptr;
  const double t_start = now_ms();
  // family 2: CommitGens from a label, as nova-snark makes them (PublicParams::setup, src/nova/proof.rs:236)
  static const char GENS_LABEL[] = "vdf-nova-ivc-v1 gens";
  auto make_gens = [&](int curve, size_t start, size_t n, vdf_bases** b) {
    return gens_family == VDF_GENS_LABEL_SHAKE
               ? vdf_bases_generate_label(ctx, curve, (const uint8_t*)GENS_LABEL, sizeof(GENS_LABEL) - 1, start, n, b)
               : vdf_bases_generate_family(ctx, curve, gens_family, GENS_SEED, start, n, b);
  };
  std::unique_ptr<vdf_pp, void (*)(vdf_pp*)> pp(new vdf_pp(), vdf_nova_pp_free);
  pp->ctx = ctx;
  pp->t = t;
  pp->circuit_kind = circuit_kind;
  pp->gens_family = gens_family;
  pp->arity = custom ? custom->arity : 3;
  pp->tune = tune;
  pp->ro = ro ? ro : ro_default();
  double* ms = pp->setup_ms;                       // [0] shapes + digest, [1] shapes to the device, [2] generators, [3] tables, [4] digit tables
  double mark = t_start;
  auto lap = [&](int k) -> int { HIPCALL(ctx, vdf_ctx_sync(ctx)); const double now = now_ms(); ms[k] += now - mark; mark = now; return VDF_OK; };
  HostShape sh[2];
  { int rc = build_shapes(t, circuit_kind, sh, custom, pp->ro); if (rc != VDF_OK) return rc; }
  digest_shapes(t, gens_family, sh, pp->digest, pp->ro);
  // only the MinRoot rounds are made on the device; a custom circuit's variables all come from the host
  pp->seg_begin = custom ? 0 : sh[PRIMARY].step_begin;
  pp->seg_len = custom ? 0 : sh[PRIMARY].step_end - sh[PRIMARY].step_begin;
  if (pp->seg_len) {
    // the longest run of primary constraints that read nothing of a witness but the segment, the step circuit's input z_in
    // (the `arity` variables allocated right before it, synthesize_augmented; known when a step begins) and the constant,
    // none of them a row the device sums by a wavefront (vdf_nifs_cross_term_rows)
    const HostShape& h = sh[PRIMARY];
    size_t best_b = 0, best_n = 0;
    longest_early_run(h, pp->seg_begin, pp->seg_len, pp->arity, &best_b, &best_n);
    if (best_n >= 64 && pp->seg_begin >= pp->arity && tune.early_rows != 0 && !(tune.flags & VDF_PP_NO_EARLY_ROWS)) { pp->ahead_row = best_b; pp->ahead_rows = best_n; }
    pp->ahead_mode = tune.early_rows == 1 ? 1 : 2;
    // the built-in circuits' early rows are a fixed stencil over the rounds' variables: compared with the shape triple by triple
    // once, here; from then on their cross term reads no sparse matrix (tuning.stencil = 0: the generic kernel, for A/B runs)
    const int per = circuit_kind == VDF_CIRCUIT_MINROOT_BOUND ? 3 : 4;
    if (!custom && pp->ahead_rows == 3 * t + 1 && tune.stencil &&
        minroot_stencil_matches(h, field(side_field(PRIMARY)), t, per, pp->seg_begin, pp->ahead_row))
      pp->stencil_per = per;
  }
  ms[0] = now_ms() - mark; mark = now_ms();
  for (int s = 0; s < 2; ++s) {
    Side& sd = pp->s[s];
    sd.side = s; sd.field = side_field(s); sd.curve = side_curve(s);
    sd.F = &field(sd.field);
    sd.Fb = &field(s == PRIMARY ? VDF_FIELD_FP : VDF_FIELD_FQ);
    sd.ctx = ctx;
    sd.arena = &pp->arena[s];
    sd.num_cons = sh[s].num_cons; sd.num_vars = sh[s].num_vars;
    sd.ncols = sd.num_vars + 1 + NUM_IO;
    memcpy(sd.digest, pp->digest, 32);
    const Coo* m = sh[s].m;
    sd.nnz3 = m[0].rows.size() + m[1].rows.size() + m[2].rows.size();
    const uint32_t* rows[3] = {m[0].rows.data(), m[1].rows.data(), m[2].rows.data()};
    const uint32_t* cols[3] = {m[0].cols.data(), m[1].cols.data(), m[2].cols.data()};
    const vdf_fe* vals[3] = {(const vdf_fe*)m[0].vals.data(), (const vdf_fe*)m[1].vals.data(), (const vdf_fe*)m[2].vals.data()};
    const size_t nnz[3] = {m[0].rows.size(), m[1].rows.size(), m[2].rows.size()};
    HIPCALL(ctx, vdf_shape_create(ctx, sd.field, sd.num_cons, sd.ncols, rows, cols, vals, nnz, &sd.shape));
    { int rc = lap(1); if (rc != VDF_OK) return rc; }
    size_t need = sd.num_vars > sd.num_cons ? sd.num_vars : sd.num_cons, g = 1;
    while (g < need) g <<= 1;                                        // next_pow2(max(vars, cons)), SURVEY.md App. C
    sd.num_gens = g;
    HIPCALL(ctx, make_gens(sd.curve, 0, g, &sd.gens));
    vdf_bases* ub = nullptr;
    HIPCALL(ctx, make_gens(sd.curve, g, 1, &ub));
    const int rc = vdf_bases_download(ctx, ub, 0, 1, (vdf_affine*)&sd.gen_u);
    vdf_bases_free(ub);
    if (rc != VDF_OK) return fail(rc, std::string("generator download: ") + vdf_last_error(ctx));
    { int rc2 = lap(2); if (rc2 != VDF_OK) return rc2; }
    // window of the fixed-base table by the size of the MSMs taken over it: 2^17 terms and more -> 16 bits, the
    // ~10^4-term witnesses of an augmented circuit -> 15 (measured: 10, 11, 13 and 15 within 4 %, 15 best)
    HIPCALL(ctx, vdf_bases_precompute(ctx, sd.gens, g >= (1u << 17) ? tune.big_window : tune.small_window, 1));
    { int rc2 = lap(3); if (rc2 != VDF_OK) return rc2; }
    HIPCALL(ctx, vdf_dev_alloc(ctx, sd.num_cons * 32, &sd.d_zero));
    HIPCALL(ctx, vdf_dev_memset(ctx, sd.d_zero, 0, sd.num_cons * 32));
    uint64_t dv[4];
    memcpy(dv, pp->digest, 32);
    pp->params[s] = int_to_fe(dv, *sd.F);
  }
  // the derived generators of the packed commitment and their table come BEFORE the optional digit tables
  if (circuit_kind == VDF_CIRCUIT_MINROOT_REFERENCE && pp->seg_len == 4 * t + 1 && tune.packed_commit) {
    mark = now_ms();
    int rc = make_packed_generators(pp.get());
    if (rc != VDF_OK) return rc;
    { int rc2 = lap(3); if (rc2 != VDF_OK) return rc2; }
  }
  // The commitments a step WAITS for are small: the secondary circuit's witness and cross term (~10^4 terms each), and
  // on the primary side what the host made of the witness and the rows of T that depend on it.  Their generators get
  // a digit table (vdf_bases_precompute_digits: a plain sum of gathered multiples, no buckets); the rounds' 2 x 10^5
  // terms, committed ahead of the step, stay with the bucket method.  Generator ranges as index intervals: the host-made
  // variables before and after the MinRoot rounds (W) and the rows of T before and after the early rows; merged where they
  // overlap (at most 4).
  mark = now_ms();
  size_t db[2][4] = {}, dn[2][4] = {}, dtot[2] = {0, 0};
  int nr[2] = {0, 0};
  for (int s = 0; s < 2; ++s) {
    const Side& sd = pp->s[s];
    const size_t top = sd.num_vars > sd.num_cons ? sd.num_vars : sd.num_cons;
    if (s == SECONDARY || pp->seg_len == 0) { dn[s][0] = top; nr[s] = 1; }
    else if (pp->ahead_rows) {
      const size_t se = pp->seg_begin + pp->seg_len, ae = pp->ahead_row + pp->ahead_rows;
      std::vector<std::pair<size_t, size_t>> iv = {{0, pp->seg_begin}, {se, sd.num_vars}, {0, pp->ahead_row}, {ae, sd.num_cons}};
      std::sort(iv.begin(), iv.end());
      std::vector<std::pair<size_t, size_t>> merged;
      for (const auto& x : iv) {
        if (x.second <= x.first) continue;
        if (!merged.empty() && x.first <= merged.back().second) merged.back().second = std::max(merged.back().second, x.second);
        else merged.push_back(x);
      }
      for (const auto& x : merged) { db[s][nr[s]] = x.first; dn[s][nr[s]] = x.second - x.first; ++nr[s]; }
    }
    for (int r = 0; r < nr[s]; ++r) dtot[s] += dn[s][r];
    if (!(nr[s] && dtot[s] <= (1u << 16))) { nr[s] = 0; dtot[s] = 0; }
  }
  // The window: what the caller fixed, or the widest of 12 .. 8 whose tables (both sides) fit the budget -- 20 GiB by default,
  // i.e. the 10-bit tables at t = 2^16 (19 GB; the 12-bit ones, 65 GB, buy ~2.5 % of a step and are for a host that says so) --
  // and the free HBM less a reserve for what comes after this call: proof buffers, three MSM workspaces, a second chain's share.
  int digit_c = (tune.flags & VDF_PP_NO_DIGIT_TABLES) ? -1 : tune.digit_window;
  if (digit_c == 0) {
    size_t free_b = 0;
    HIPCALL(ctx, vdf_dev_mem_info(ctx, &free_b, nullptr));
    const uint64_t reserve = (uint64_t)4 << 30;
    const uint64_t room = free_b > reserve ? free_b - reserve : 0;
    const uint64_t budget = tune.digit_budget_bytes < room ? tune.digit_budget_bytes : room;
    digit_c = -1;
    for (int c = 12; c >= 8; --c)
      if ((uint64_t)vdf_digit_table_bytes(c, dtot[0] + dtot[1]) <= budget) { digit_c = c; break; }
    if (tune.verbose) fprintf(stderr, "vdf_nova: digit window %d (%zu generators, budget %.1f GiB, free %.1f GiB)\n", digit_c, dtot[0] + dtot[1],
                              budget / 1073741824.0, free_b / 1073741824.0);
    if (digit_c < 0) for (int s = 0; s < 2; ++s) if (nr[s]) pp->digit_tables_skipped |= 1u << s;
  }
  for (int s = 0; s < 2 && digit_c > 0; ++s) {
    if (!nr[s]) continue;
    // never a requirement: when it does not fit the free HBM or the window is refused, the parameters are made without
    // it and vdf_msm takes the bucket method
    const int rc = vdf_bases_precompute_digits(ctx, pp->s[s].gens, digit_c, nr[s], db[s], dn[s]);
    if (rc == VDF_ERR_OOM || rc == VDF_ERR_BAD_ARG) {
      pp->digit_tables_skipped |= 1u << s;
      if (tune.verbose) fprintf(stderr, "vdf_nova: no digit table on side %d (%s): commitments take the bucket method\n", s, vdf_last_error(ctx));
    } else if (rc != VDF_OK) {
      return fail(rc, std::string("vdf_bases_precompute_digits: ") + vdf_last_error(ctx));
    } else {
      pp->digit_table_bytes[s] = vdf_bases_digit_table_bytes(pp->s[s].gens);
    }
  }
  { int rc2 = lap(4); if (rc2 != VDF_OK) return rc2; }
  ms[6] = now_ms() - t_start;
  ms[5] = ms[6] - (ms[0] + ms[1] + ms[2] + ms[3] + ms[4]);
  *out = pp.release();
  return VDF_OK;
}
void vdf_nova_pp_free(vdf_pp* pp) {
  if (!pp) return;
  if (pp->aux_ctx) vdf_ctx_destroy(pp->aux_ctx);
  if (pp->aux_ctx2) vdf_ctx_destroy(pp->aux_ctx2);
  for (auto& a : pp->arena) if (a.p) vdf_dev_free(pp->ctx, a.p);
  if (pp->seg_gens) vdf_bases_free(pp->seg_gens);
  for (Side& sd : pp->s) {
    if (sd.d_zero) vdf_dev_free(pp->ctx, sd.d_zero);
    if (sd.shape) vdf_shape_free(sd.shape);
    if (sd.gens) vdf_bases_free(sd.gens);
  }
  delete pp;
}
int vdf_nova_pp_sizes(const vdf_pp* pp, int side, uint64_t* num_cons, uint64_t* num_vars, uint64_t* num_io, uint64_t* nnz3,
                      uint64_t* num_gens) {
  if (!pp || (side != PRIMARY && side != SECONDARY)) return fail(VDF_ERR_BAD_ARG, "bad argument");
  const Side& sd = pp->s[side];
  if (num_cons) *num_cons = sd.num_cons;
  if (num_vars) *num_vars = sd.num_vars;
  if (num_io) *num_io = NUM_IO;
  if (nnz3) *nnz3 = sd.nnz3;
  if (num_gens) *num_gens = sd.num_gens;
  return VDF_OK;
}
int vdf_nova_pp_memory(const vdf_pp* pp, uint64_t gens_bytes[2], uint64_t table_bytes[2], uint64_t digit_bytes[2], uint32_t* skipped) {
  if (!pp) return fail(VDF_ERR_BAD_ARG, "null argument");
  for (int s = 0; s < 2; ++s) {
    const Side& sd = pp->s[s];
    // sizes as the library holds them (tables x generators x 64 B); the primary side includes the packed commitment's derived generators
    const vdf_bases* extra = s == PRIMARY ? pp->seg_gens : nullptr;
    if (gens_bytes) gens_bytes[s] = ((uint64_t)sd.num_gens + (extra ? vdf_bases_len(extra) : 0)) * 64;
    if (table_bytes) table_bytes[s] = (uint64_t)vdf_bases_table_bytes(sd.gens) + (extra ? vdf_bases_table_bytes(extra) : 0);
    if (digit_bytes) digit_bytes[s] = pp->digit_table_bytes[s];
  }
  if (skipped) *skipped = pp->digit_tables_skipped;
  return VDF_OK;
}
int vdf_nova_pp_digest(const vdf_pp* pp, uint8_t out[32]) {
  if (!pp || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
  memcpy(out, pp->digest, 32);
  return VDF_OK;
}
int vdf_nova_pp_segment(const vdf_pp* pp, uint64_t* begin, uint64_t* len) {
  if (!pp) return fail(VDF_ERR_BAD_ARG, "null argument");
  if (begin) *begin = pp->seg_begin;
  if (len) *len = pp->seg_len;
  return VDF_OK;
}
int vdf_nova_pp_early_rows(const vdf_pp* pp, uint64_t* begin, uint64_t* len) {
  if (!pp) return fail(VDF_ERR_BAD_ARG, "null argument");
  if (begin) *begin = pp->ahead_row;
  if (len) *len = pp->ahead_rows;
  return VDF_OK;
}
int vdf_nova_pp_stencil(const vdf_pp* pp) { return pp ? pp->stencil_per : 0; }
// host only (no device): what vdf_nova_public_params would find for the built-in step circuit `circuit_kind` at `t`
int vdf_nova_shape_stencil(uint64_t t, int circuit_kind, uint64_t* early_begin, uint64_t* early_len, uint64_t* seg_begin) {
  return nova_guard([&]() -> int {
    if (t == 0 || t > (1ull << 24) || (circuit_kind != VDF_CIRCUIT_MINROOT_BOUND && circuit_kind != VDF_CIRCUIT_MINROOT_REFERENCE))
      return -fail(VDF_ERR_BAD_ARG, "bad argument");
    HostShape sh[2];
    if (build_shapes(t, circuit_kind, sh) != VDF_OK) return -VDF_ERR_DEVICE;
    const HostShape& h = sh[PRIMARY];
    const size_t sb = h.step_begin, sl = h.step_end - h.step_begin;
    size_t b = 0, n = 0;
    if (sb < 3) return 0;
    longest_early_run(h, sb, sl, 3, &b, &n);
    if (early_begin) *early_begin = b;
    if (early_len) *early_len = n;
    if (seg_begin) *seg_begin = sb;
    const int per = circuit_kind == VDF_CIRCUIT_MINROOT_BOUND ? 3 : 4;
    return (n == 3 * t + 1 && minroot_stencil_matches(h, field(side_field(PRIMARY)), t, per, sb, b)) ? per : 0;
  });
}

// ---- circuits ----------------------------------------------------------------------------------------
int vdf_nova_eval_and_make_circuits(int mode, uint64_t t, size_t num_steps, const vdf_state* initial_state,
                                    vdf_fe z0_primary[3], vdf_circuits** out) {
  return nova_guard([&]() -> int {
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
      vdf_minroot_eval(VDF_FIELD_FQ, mode, &in, t, &res, (vdf_fe*)c.trace_xy.data());
      c.result = load_state(&res);
      state = c.result;
      cs->v.push_back(std::move(c));
    }
    memcpy(z0_primary, &state, 96);                                    // z0 = final state, :278-281
    std::vector<Circuit> rev(cs->v.rbegin(), cs->v.rend());            // circuits.reverse(), :294
    cs->v.swap(rev);
    *out = cs;
    return VDF_OK;
  });
}
int vdf_nova_circuits_upload(vdf_ctx* ctx, vdf_circuits* c) {
  return nova_guard([&]() -> int {
    if (!ctx || !c) return fail(VDF_ERR_BAD_ARG, "null argument");
    c->ctx = ctx;
    for (auto& k : c->v) {
      if (k.d_trace) continue;
      HIPCALL(ctx, vdf_dev_alloc(ctx, k.trace_xy.size() * 32, &k.d_trace));
      HIPCALL(ctx, vdf_dev_memcpy(ctx, k.d_trace, k.trace_xy.data(), k.trace_xy.size() * 32));
    }
    return VDF_OK;
  });
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
static int prove_step_impl(vdf_pp* pp, vdf_proof** proof, const vdf_circuits* circuits, size_t k, const vdf_step_circuit* custom,
                           const vdf_fe* z0, vdf_proof** fresh);

int vdf_nova_prove_step(vdf_pp* pp, vdf_proof** proof, const vdf_circuits* circuits, size_t k, const vdf_fe z0[3]) {
  if (pp && pp->circuit_kind == VDF_CIRCUIT_CUSTOM) return fail(VDF_ERR_BAD_ARG, "these parameters are for a custom step circuit: vdf_nova_prove_step_custom");
  vdf_proof* fresh = nullptr;                     // a proof object this call created (the `None` case)
  int rc;
  try { rc = prove_step_impl(pp, proof, circuits, k, nullptr, z0, &fresh); }
  catch (const std::exception& ex) { rc = fail(VDF_ERR_DEVICE, ex.what()); }
  if (rc != VDF_OK && fresh) vdf_nova_proof_free(fresh);      // never leak a half-built proof
  return rc;
}

int vdf_nova_prove_step_custom(vdf_pp* pp, vdf_proof** proof, const vdf_step_circuit* primary, const vdf_fe* z0) {
  if (!pp || !primary || !primary->synthesize) return fail(VDF_ERR_BAD_ARG, "null argument");
  if (pp->circuit_kind != VDF_CIRCUIT_CUSTOM || primary->arity != pp->arity)
    return fail(VDF_ERR_BAD_ARG, "the circuit does not belong to these parameters");
  vdf_proof* fresh = nullptr;
  int rc;
  try { rc = prove_step_impl(pp, proof, nullptr, 0, primary, z0, &fresh); }
  catch (const std::exception& ex) { rc = fail(VDF_ERR_DEVICE, ex.what()); }
  if (rc != VDF_OK && fresh) vdf_nova_proof_free(fresh);
  return rc;
}

// host-synthesised part of a witness into its device vector [W | 1 | X] through pinned memory; the variables
// [cs.dev_begin, cs.dev_begin + cs.dev_len) are already there (the GPU made them), cs.W holds the others packed
static int upload_fresh(vdf_ctx* ctx, const Side& sd, const CS& cs, Fe* stage, void* d_z) {
  const size_t nv = sd.num_vars, nh = cs.W.size();
  if (cs.num_vars() != nv || cs.rows != sd.num_cons) return fail(VDF_ERR_DEVICE, "augmented circuit does not match its shape");
  memcpy(stage, cs.W.data(), nh * 32);
  stage[nh] = one(*sd.F);
  stage[nh + 1] = cs.X[0]; stage[nh + 2] = cs.X[1];
  if (cs.dev_len == 0) {
    HIPCALL(ctx, vdf_dev_memcpy(ctx, d_z, stage, (nh + 3) * 32));
  } else {
    const size_t b = cs.dev_begin, e = cs.dev_begin + cs.dev_len;
    if (b) HIPCALL(ctx, vdf_dev_memcpy(ctx, d_z, stage, b * 32));
    HIPCALL(ctx, vdf_dev_memcpy(ctx, (char*)d_z + e * 32, stage + b, (nh + 3 - b) * 32));
  }
  return VDF_OK;
}

// ---- prove_step ----------------------------------------------------------------------------------------------
// One IVC step drives THREE device queues and the host between them (StepRun below: one method per phase, grouped by the
// queue it feeds).  The programmes, and the marks (vdf_ctx_mark slots) through which they meet:
//
//   CHAIN       the caller's context (pp->ctx) and the calling thread: NIFS of the secondary side (a), synthesis of the
//               primary circuit (b), NIFS of the primary side (c), synthesis of the secondary circuit and the folds (d) --
//               a step's critical path.
//   LOOKAHEAD   p->ctx2[j % DEPTH]: the MinRoot rounds of step k + 1 from the forward trace, and their commitment.
//   EARLY ROWS  p->ctx3: the rows of the primary cross term T that read only those rounds, and their commitment.
//
//   MARK_Z        set on LOOKAHEAD   the rounds of a ring slot are written            EARLY ROWS start behind it; a step ends behind it
//   MARK_W        set on LOOKAHEAD   their commitment has landed in h_pts[slot]       CHAIN (c) waits for it before it sums comm_W
//   MARK_T        set on EARLY ROWS  the early rows' commitment has landed in hb[4]   CHAIN (c) waits for it before it sums comm_T
//   MARK_STEP     set on CHAIN       the step's uploads have left the staging buffers  the call returns behind it
//   MARK_PRIMARY  set on CHAIN       the primary side's direct sums are enqueued      the LOOKAHEAD's bucket accumulation is gated on it
//   MARK_FOLD     set on the fold's queue (EARLY ROWS, or CHAIN)                      the next step's EARLY ROWS start behind it; CHAIN waits for it at the end
//   MARK_ZIN      set on CHAIN       the next step's z_in is in its ring slot         the fold on the EARLY ROWS queue waits for it
// (slots 4..7 are the ones include/vdf_hip.h keeps for this library on the caller's context)
namespace {
enum StepMark { MARK_Z = 0, MARK_W = 1, MARK_T = 2, MARK_STEP = 4, MARK_PRIMARY = 5, MARK_FOLD = 6, MARK_ZIN = 7 };

struct StepRun {
  static constexpr int D = vdf_proof::DEPTH, R = vdf_proof::RING;
  // ---- what the step works on
  vdf_pp* const pp;
  vdf_proof* const p;
  const vdf_circuits* const circuits;
  const size_t k;
  const vdf_step_circuit* const custom;
  const Circuit& c;
  const bool first;
  const size_t arity;
  vdf_ctx* const ctx;                  // CHAIN
  vdf_ctx* const ct;                   // EARLY ROWS
  const Side& S1;
  const Side& S2;
  const Field& F1;
  const Field& F2;
  const size_t seg_b, seg_n, seg_e;    // the primary witness's run of round variables
  const int per;
  const bool t_ahead;                  // this step has early rows
  const size_t ta_b, ta_n, ta_e;
  const int t_parts;                   // (tuning.early_row_parts = 2 or 3: the early rows as an MSM job of that many parts (vdf_msm_job_*): a later part's
                                       // rows and sort run under an earlier part's bucket accumulation, one shared bucket reduction.  r3: 1.10 ms per step against 0.95 as ONE MSM, the default)
  const bool fold_on_rows;             // (tuning.fold_on_rows = 0: the primary fold on the main queue, the early rows waiting for its mark)
  vdf_jac* const hb;                   // four result slots of the batched commitments, then the early rows' parts
  // ---- state of this step
  double t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
  bool hit = false;                    // the rounds of this step were made by the previous step's lookahead
  int slot = 0;                        // ring slot of this step's fresh primary witness
  vdf_ctx* cq = nullptr;               // LOOKAHEAD queue that made this step's rounds
  void* d_z2 = nullptr;
  int zin_slot = -1;                   // ring slot whose z_in this step has already uploaded (for the next step's early rows)
  bool fold_elsewhere = false;         // the primary fold ran on the early rows' queue: the main queue waits for it before the step ends
  struct { bool pending = false; void* d_next = nullptr; vdf_ctx* cq_next = nullptr; vdf_ctx* fq = nullptr; int slot = -1; bool fused = false; Fe fold_r, u_folded; } deferred;   // the next step's early rows
  bool gate_next_segment = false;      // the next lookahead_enqueue holds its bucket accumulation behind MARK_PRIMARY
  bool touched[D] = {};
  bool looked = false, waited_w = false;
  Aff comm_T2, comm_T1;
  uint64_t r2[4] = {0, 0, 0, 0}, r1[4] = {0, 0, 0, 0};
  AugInputs in1, in2;
  std::unique_ptr<StepCircuit> c1;
  const TrivialTestCircuit c2;
  AugEarlyPtr early1, early2;
  Inst l1;

  StepRun(vdf_pp* pp_, vdf_proof* p_, const vdf_circuits* circuits_, size_t k_, const vdf_step_circuit* custom_, const Circuit& c_, bool first_)
      : pp(pp_), p(p_), circuits(circuits_), k(k_), custom(custom_), c(c_), first(first_), arity(pp_->arity), ctx(pp_->ctx), ct(p_->ctx3),
        S1(pp_->s[PRIMARY]), S2(pp_->s[SECONDARY]), F1(*pp_->s[PRIMARY].F), F2(*pp_->s[SECONDARY].F), seg_b(pp_->seg_begin), seg_n(pp_->seg_len),
        seg_e(pp_->seg_begin + pp_->seg_len), per(pp_->circuit_kind == VDF_CIRCUIT_MINROOT_BOUND ? 3 : 4),
        t_ahead(!first_ && !custom_ && pp_->ahead_rows != 0), ta_b(pp_->ahead_row), ta_n(pp_->ahead_rows), ta_e(pp_->ahead_row + pp_->ahead_rows),
        t_parts(pp_->tune.early_row_parts), fold_on_rows(pp_->tune.fold_on_rows != 0), hb(&p_->h_pts[R]), early1(nullptr, aug_early_free),
        early2(nullptr, aug_early_free) {
    memset(&comm_T2, 0, sizeof(Aff)); memset(&comm_T1, 0, sizeof(Aff));
  }

  // ================================ LOOKAHEAD queue ==================================================================
  // the MinRoot rounds of step j into ring slot s on that step's lookahead context, and their share of the commitment
  int lookahead_enqueue(size_t j, int s, bool cold) {
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
    char* seg = (char*)p->d_z2s[s] + seg_b * 32;
    void* packed = pp->seg_gens ? p->d_packed[j % D] : nullptr;
    if (packed) HIPCALL(q, vdf_minroot_step_segment_packed(q, S1.field, (const vdf_fe*)d_trace, pp->t, (const vdf_fe*)&cc.input.i,
                                                            (const vdf_fe*)&cc.result.i, (vdf_fe*)seg, (vdf_fe*)packed));
    else HIPCALL(q, vdf_minroot_step_segment(q, S1.field, (const vdf_fe*)d_trace, pp->t, (const vdf_fe*)&cc.input.i, per, (vdf_fe*)seg));
    HIPCALL(q, vdf_ctx_mark(q, MARK_Z));
    HIPCALL(ctx, vdf_ctx_wait(ctx, q));          // the first context waits for the rounds only, not for their commitment
    // the reference's circuit: the same commitment from 3t + 4 terms over the derived generators (make_packed_generators)
    if (gate_next_segment) { HIPCALL(q, vdf_ctx_gate_accumulate(q, ctx, MARK_PRIMARY)); gate_next_segment = false; }
    if (packed) HIPCALL(q, vdf_msm(q, pp->seg_gens, 0, (const vdf_fe*)packed, 3 * pp->t + 4, 1, &p->h_pts[s]));
    else HIPCALL(q, vdf_msm(q, S1.gens, seg_b, (const vdf_fe*)seg, seg_n, 1, &p->h_pts[s]));
    HIPCALL(q, vdf_ctx_mark(q, MARK_W));
    touched[j % D] = true;
    vdf_proof::Ahead a;
    a.result = cc.result; a.input = cc.input; a.slot = s;
    p->ahead.push_back(a);
    return VDF_OK;
  }
  // which ring slot holds this step's rounds: the previous step's lookahead (a hit), or made now
  int lookahead_select() {
    hit = !custom && !p->ahead.empty() && p->ahead_circuits == circuits && p->ahead_k == k &&
          memcmp(&p->ahead[0].result, &c.result, sizeof(St)) == 0 && memcmp(&p->ahead[0].input, &c.input, sizeof(St)) == 0;
    if (custom) {
      // every variable comes from the host: no rounds to make ahead of time, the ring just rotates
      vdf_proof::Ahead a;
      a.slot = first ? 0 : (p->slot + 1) % R;
      p->ahead.assign(1, a);
    } else if (!hit) {
      if (!p->ahead.empty()) for (vdf_ctx* q : p->ctx2) HIPCALL(q, vdf_ctx_sync(q));       // lookaheads nobody came for
      p->ahead.clear();
      p->ahead_circuits = circuits;
      p->ahead_k = k;
      int rc = lookahead_enqueue(k, first ? 0 : (p->slot + 1) % R, !first);
      if (rc != VDF_OK) return rc;
    }
    slot = p->ahead[0].slot;
    cq = p->ctx2[k % D];
    p->slot = slot;
    d_z2 = p->d_z2s[slot];
    return VDF_OK;
  }
  // this step's entry leaves the list; the rounds of the steps that follow are enqueued
  int lookahead_advance() {
    p->ahead.erase(p->ahead.begin());
    if (custom) return VDF_OK;
    p->ahead_k = k + 1;
    while (p->ahead.size() < (size_t)D) {
      const size_t j = p->ahead_k + p->ahead.size();
      if (j >= circuits->v.size() || circuits->v[j].t != pp->t) break;
      const int last = p->ahead.empty() ? slot : p->ahead.back().slot;
      int rc = lookahead_enqueue(j, (last + 1) % R, !hit);
      if (rc != VDF_OK) return rc;
    }
    return VDF_OK;
  }

  // ================================ EARLY ROWS queue =================================================================
  // The constraints of the MinRoot rounds read nothing of a step's witness but the rounds themselves, which are on the
  // device already, and the running instance they are crossed with is final since the previous fold: their rows of T, and
  // that part of comm_T, run on a queue of their own beside the secondary side's NIFS and the host's synthesis of the
  // primary circuit, and leave ~10^4 rows instead of 2 x 10^5 on the critical path.  For the step whose fresh witness
  // lives in d_z2 (this step's, or -- launched on the way out -- the next one's).
  // fold_r (fused mode, stencil_fold_ok()): the challenge of the fold that has NOT been applied to the early rows of the running
  // A z, B z, C z and E yet -- the stencil applies it on the way (vdf_nifs_cross_term_minroot_fold) and MARK_FOLD is set behind it
  bool stencil_fold_ok() const { return pp->tune.fold_fused != 0 && pp->stencil_per != 0 && (t_parts == 1 || ta_n < 4096); }
  int early_rows_launch(void* d_z2, vdf_ctx* cq, bool zin_in_place, const Fe* fold_r = nullptr) {
    SideState& s1 = p->r[PRIMARY];
    HIPCALL(ct, vdf_ctx_wait_mark(ct, cq, MARK_Z));                 // the rounds are in place (written a step ago, normally)
    // z_in = z_i past the base step (the circuit's selection); the same values arrive again with the host's variables
    if (!zin_in_place) {
      memcpy(p->h_zin, p->zi[PRIMARY].data(), arity * 32);          // pinned: the copy reads it when it runs
      HIPCALL(ct, vdf_dev_memcpy(ct, (char*)d_z2 + (seg_b - arity) * 32, p->h_zin, arity * 32));
    }
    auto rows = [&](size_t b, size_t n) -> int {
      if (fold_r) {                                                  // the stencil with the previous fold of its rows on the way
        if (!(pp->stencil_per && b == ta_b && n == ta_n)) return fail(VDF_ERR_DEVICE, "fused fold without the stencil");
        HIPCALL(ct, vdf_nifs_cross_term_minroot_fold(ct, S1.field, pp->stencil_per, pp->t, seg_b, S1.num_vars, b, (const vdf_fe*)d_z2,
                                                     (const vdf_fe*)fold_r, (vdf_fe*)s1.d_abc[0], (vdf_fe*)s1.d_abc[1], (vdf_fe*)s1.d_abc[2],
                                                     (vdf_fe*)s1.d_E, (const vdf_fe*)s1.d_T, (const vdf_fe*)&s1.inst.u, (vdf_fe*)s1.d_abc2[0],
                                                     (vdf_fe*)s1.d_abc2[1], (vdf_fe*)s1.d_abc2[2], (vdf_fe*)s1.d_T));
        HIPCALL(ct, vdf_ctx_mark(ct, MARK_FOLD));                    // the running instance is whole again from here on
        return VDF_OK;
      }
      if (pp->stencil_per && b == ta_b && n == ta_n) {               // the MinRoot stencil: streams only (vdf_hip.h)
        HIPCALL(ct, vdf_nifs_cross_term_minroot(ct, S1.field, pp->stencil_per, pp->t, seg_b, S1.num_vars, b, (const vdf_fe*)d_z2,
                                                (const vdf_fe*)s1.d_abc[0], (const vdf_fe*)s1.d_abc[1], (const vdf_fe*)s1.d_abc[2],
                                                (const vdf_fe*)&s1.inst.u, (vdf_fe*)s1.d_abc2[0], (vdf_fe*)s1.d_abc2[1],
                                                (vdf_fe*)s1.d_abc2[2], (vdf_fe*)s1.d_T));
        return VDF_OK;
      }
      HIPCALL(ct, vdf_nifs_cross_term_rows(ct, S1.shape, b, n, VDF_ROWS_INSIDE, (const vdf_fe*)d_z2, (const vdf_fe*)s1.d_abc[0],
                                           (const vdf_fe*)s1.d_abc[1], (const vdf_fe*)s1.d_abc[2], (const vdf_fe*)&s1.inst.u,
                                           (vdf_fe*)s1.d_abc2[0], (vdf_fe*)s1.d_abc2[1], (vdf_fe*)s1.d_abc2[2], (vdf_fe*)s1.d_T));
      return VDF_OK;
    };
    memset(&hb[5], 0, 2 * sizeof(vdf_jac));                          // parts not used this time: the identity
    if (t_parts == 1 || ta_n < 4096) {
      int rc = rows(ta_b, ta_n);
      if (rc != VDF_OK) return rc;
      HIPCALL(ct, vdf_msm(ct, S1.gens, ta_b, (const vdf_fe*)((const char*)s1.d_T + ta_b * 32), ta_n, 1, &hb[4]));
    } else {
      size_t off[3], len[3];
      for (int g = 0; g < t_parts; ++g) { off[g] = ta_b + ta_n * g / t_parts; len[g] = ta_b + ta_n * (g + 1) / t_parts - off[g]; }
      vdf_msm_job* job = nullptr;
      HIPCALL(ct, vdf_msm_job_begin(ct, S1.gens, t_parts, off, len, 1, &job));
      for (int g = 0; g < t_parts; ++g) {
        int rc = rows(off[g], len[g]);
        if (rc == VDF_OK && vdf_msm_job_push(job, g, (const vdf_fe*)((const char*)s1.d_T + off[g] * 32)) != VDF_OK)
          rc = fail(VDF_ERR_DEVICE, std::string("vdf_msm_job_push: ") + vdf_last_error(ct));
        if (rc != VDF_OK) { vdf_jac scratch[3]; (void)vdf_msm_job_finish(job, scratch); return rc; }
      }
      HIPCALL(ct, vdf_msm_job_finish(job, &hb[4]));
    }
    HIPCALL(ct, vdf_ctx_mark(ct, MARK_T));
    return VDF_OK;
  }

  // ================================ CHAIN: the caller's queue and the calling thread ================================
  // the two circuits' inputs but for the commitments the device is still making; the primary step circuit
  void chain_prepare_inputs() {
    in1.ro = pp->ro;
    in1.params = pp->params[PRIMARY];
    in1.i = from_u64((uint64_t)p->i, F1);
    in1.z0 = p->z0[PRIMARY];
    in1.zi = p->zi[PRIMARY];
    if (first) {
      const AugInputs b = blank_inputs(arity);
      in1.U = b.U; in1.u_W = b.u_W; memcpy(in1.u_X, b.u_X, sizeof(in1.u_X)); in1.T = b.T;
    } else {
      in1.U = to_relaxed(p->r[SECONDARY].inst, F2);
      memset(&in1.u_W, 0, sizeof(Aff)); memset(&in1.T, 0, sizeof(Aff));
      for (int j = 0; j < 2; ++j) fe_to_int(p->l2.X[j], F2, in1.u_X[j]);
    }
    c1 = custom ? make_custom_circuit(custom) : make_primary_circuit(pp, &c, true);
    in2.ro = pp->ro;
  }
  // cross term of (running secondary, l2), commit(w2) unless known, commit(T2)
  int chain_launch_nifs2() {
    SideState& s2 = p->r[SECONDARY];
    HIPCALL(ctx, vdf_nifs_cross_term(ctx, S2.shape, (const vdf_fe*)p->d_l2z, (const vdf_fe*)s2.d_abc[0], (const vdf_fe*)s2.d_abc[1],
                                     (const vdf_fe*)s2.d_abc[2], (const vdf_fe*)&s2.inst.u, (vdf_fe*)s2.d_abc2[0], (vdf_fe*)s2.d_abc2[1],
                                     (vdf_fe*)s2.d_abc2[2], (vdf_fe*)s2.d_T));
    if (p->l2_committed) {
      HIPCALL(ctx, vdf_msm(ctx, S2.gens, 0, (const vdf_fe*)s2.d_T, S2.num_cons, 1, &hb[1]));
    } else {
      const size_t off[2] = {0, 0}, len[2] = {S2.num_vars, S2.num_cons};
      const vdf_fe* sc[2] = {(const vdf_fe*)p->d_l2z, (const vdf_fe*)s2.d_T};
      HIPCALL(ctx, vdf_msm_batch(ctx, S2.gens, 2, off, sc, len, 1, hb));
    }
    return VDF_OK;
  }
  // ---- (a) NIFS on the secondary side: cross term of (running, l2), commitments of l2's witness and of T ----------
  int chain_secondary_nifs() {
    if (!first) {
      if (p->nifs2 == vdf_proof::NIFS2_NONE) {       // normally in flight since the previous step's last lines
        int rc = chain_launch_nifs2();
        if (rc != VDF_OK) return rc;
        p->nifs2 = vdf_proof::NIFS2_INFLIGHT;
      }
      // launched while the host waits for this side's commitments, on a queue of their own: their 0.6 ms must be over when
      // the primary side's own commitments are (0.65 ms into the step), and this side's direct sum, one prioritised
      // wavefront per SIMD for 0.1 ms, loses little to a bucket accumulation beside it
      const bool rows_inflight = t_ahead && hit && p->tahead_valid && p->tahead_slot == slot && p->tahead_k == k &&
                                 p->tahead_circuits == circuits;       // launched by the previous step on its way out
      if (t_ahead && !rows_inflight) {
        if (pp->ahead_mode == 1) HIPCALL(ct, vdf_ctx_wait(ct, ctx));
        else if (p->tahead_valid) HIPCALL(ct, vdf_ctx_sync(ct));       // rows made for a step that did not come: let them finish
        int rc = early_rows_launch(d_z2, cq, false);
        if (rc != VDF_OK) return rc;
      }
      p->tahead_valid = false;
      early1 = synthesize_augmented_early(PRIMARY, in1, *c1);      // the host's share of the wait: what the circuit can do without T
      { int rc = finalize_l2(p); if (rc != VDF_OK) return rc; }      // waits, collects comm_W2 and comm_T2
      comm_T2 = p->nifs2_T;
      p->nifs2 = vdf_proof::NIFS2_NONE;               // consumed: the fold below uses the scratch vectors up
    }
    t1 = now_ms();
    return VDF_OK;
  }
  // ---- (b) the primary augmented circuit -----------------------------------------------------------------------
  int chain_primary_circuit() {
    {
      AugInputs& in = in1;
      if (!first) { in.u_W = p->l2.comm_W; in.T = comm_T2; }
      CS cs(S1.field, false, pp->ro);
      Fe unew[9];
      const std::vector<Fe> z_next = synthesize_augmented(cs, PRIMARY, in, *c1, unew, r2, early1.get());
      early1.reset();
      if (custom && static_cast<const CustomStepCircuit*>(c1.get())->rc != 0) return fail(VDF_ERR_BAD_ARG, "the step circuit's synthesize failed");
      if (cs.dev_len != seg_n || (seg_n && cs.dev_begin != seg_b)) return fail(VDF_ERR_DEVICE, "device segment moved");
      t2 = now_ms();
      p->r[SECONDARY].inst = inst_from_elements(unew, F1, F2);       // base step: the default instance
      // the host-made variables go next to the rounds the lookahead context has written (vdf_ctx_wait at their launch)
      int rc = upload_fresh(ctx, S1, cs, p->h_stage[PRIMARY], d_z2);
      if (rc != VDF_OK) return rc;
      if (!first && !custom && pp->ahead_rows != 0 && seg_n) {
        // the NEXT step's z_in = this step's output, known now: into its place in the next ring slot's fresh witness, half a
        // step before the early rows of that step read it
        zin_slot = (slot + 1) % R;
        memcpy(p->h_zin + arity, z_next.data(), arity * 32);          // the second pinned slot (the first may still feed this step's early rows)
        HIPCALL(ctx, vdf_dev_memcpy(ctx, (char*)p->d_z2s[zin_slot] + (seg_b - arity) * 32, p->h_zin + arity, arity * 32));
        HIPCALL(ctx, vdf_ctx_mark(ctx, MARK_ZIN));
      }
      l1.X[0] = cs.X[0]; l1.X[1] = cs.X[1];
      l1.u = one(F1);
      memset(&l1.comm_E, 0, sizeof(Aff));
      p->zi[PRIMARY] = z_next;
    }
    return VDF_OK;
  }
  // ---- (c) NIFS on the primary side ----------------------------------------------------------------------------
  int chain_primary_nifs() {
    {
      SideState& s1 = p->r[PRIMARY];
      // host-made variables before and after the device's run (one group when there is no such run), then T -- all of it,
      // or the rows before and after the ones committed early
      size_t off[4], len[4];
      const vdf_fe* sc[4];
      int ng = 0, first_T = 0;
      auto group = [&](const void* base, size_t begin, size_t n) {
        if (n == 0) return;
        off[ng] = begin; len[ng] = n; sc[ng] = (const vdf_fe*)((const char*)base + begin * 32); ++ng;
      };
      if (seg_n) { group(d_z2, 0, seg_b); group(d_z2, seg_e, S1.num_vars - seg_e); }
      else group(d_z2, 0, S1.num_vars);
      first_T = ng;
      if (!first) {
        if (t_ahead) {
          group(s1.d_T, 0, ta_b); group(s1.d_T, ta_e, S1.num_cons - ta_e);
          HIPCALL(ctx, vdf_nifs_cross_term_rows(ctx, S1.shape, ta_b, ta_n, VDF_ROWS_OUTSIDE, (const vdf_fe*)d_z2, (const vdf_fe*)s1.d_abc[0],
                                                (const vdf_fe*)s1.d_abc[1], (const vdf_fe*)s1.d_abc[2], (const vdf_fe*)&s1.inst.u,
                                                (vdf_fe*)s1.d_abc2[0], (vdf_fe*)s1.d_abc2[1], (vdf_fe*)s1.d_abc2[2], (vdf_fe*)s1.d_T));
        } else {
          group(s1.d_T, 0, S1.num_cons);
          HIPCALL(ctx, vdf_nifs_cross_term(ctx, S1.shape, (const vdf_fe*)d_z2, (const vdf_fe*)s1.d_abc[0], (const vdf_fe*)s1.d_abc[1],
                                           (const vdf_fe*)s1.d_abc[2], (const vdf_fe*)&s1.inst.u, (vdf_fe*)s1.d_abc2[0], (vdf_fe*)s1.d_abc2[1],
                                           (vdf_fe*)s1.d_abc2[2], (vdf_fe*)s1.d_T));
        }
      }
      HIPCALL(ctx, vdf_msm_batch(ctx, S1.gens, ng, off, sc, len, 1, hb));
      // the lookahead launched under this wait sorts beside these commitments but holds its bucket accumulation -- every SIMD
      // for 0.25 ms -- until they are done: it then runs while the host synthesises the secondary circuit (tuning.gate_accumulate = 0: no hold)
      const bool gate = pp->tune.gate_accumulate != 0;
      if (gate && !first && !custom) { HIPCALL(ctx, vdf_ctx_mark(ctx, MARK_PRIMARY)); gate_next_segment = true; }
      if (!first) {
        // behind the primary side's launches (it does not depend on them, they do not wait for it): the fold of the secondary
        // witness on the device (z, E, A z, B z, C z += r2 * fresh); the instance came from the circuit
        SideState& s2 = p->r[SECONDARY];
        const Fe rr = int_to_fe(r2, F2);
        vdf_fe* acc[5] = {(vdf_fe*)s2.d_z, (vdf_fe*)s2.d_E, (vdf_fe*)s2.d_abc[0], (vdf_fe*)s2.d_abc[1], (vdf_fe*)s2.d_abc[2]};
        const vdf_fe* addv[5] = {(const vdf_fe*)p->d_l2z, (const vdf_fe*)s2.d_T, (const vdf_fe*)s2.d_abc2[0], (const vdf_fe*)s2.d_abc2[1],
                                 (const vdf_fe*)s2.d_abc2[2]};
        const size_t len[5] = {S2.ncols, S2.num_cons, S2.num_cons, S2.num_cons, S2.num_cons};
        HIPCALL(ctx, vdf_fold_many(ctx, S2.field, (const vdf_fe*)&rr, 5, acc, addv, len));
      }
      t3 = now_ms();
      // The next step's MinRoot rounds and their commitment go to the second queue NOW, under this wait: their dozen launches
      // cost the chain nothing here, and their 0.65 ms are over that much sooner (they are the next step's primary
      // commitment; back to back the device is the co-bottleneck).  This step's own rounds were committed a step ago; their
      // mark is waited for first, because the launch reuses it.  (tuning.lookahead_early = 0: launched after the wait.)
      const bool la_early = pp->tune.lookahead_early != 0;
      if (la_early && !first && !custom) {
        if (seg_n) HIPCALL(cq, vdf_ctx_sync_mark(cq, MARK_W));
        waited_w = true;
        int rc = lookahead_advance();
        if (rc != VDF_OK) return rc;
        looked = true;
      }
      // the host's share of this wait: the secondary circuit's inputs but for comm_W and comm_T, and what it can do with them
      in2.params = pp->params[SECONDARY];
      in2.i = from_u64((uint64_t)p->i, F2);
      in2.z0 = p->z0[SECONDARY];
      in2.zi = p->zi[SECONDARY];
      if (first) { const AugInputs b = blank_inputs(1); in2.U = b.U; }
      else in2.U = to_relaxed(p->r[PRIMARY].inst, F1);
      memset(&in2.u_W, 0, sizeof(Aff)); memset(&in2.T, 0, sizeof(Aff));
      for (int j = 0; j < 2; ++j) fe_to_int(l1.X[j], F1, in2.u_X[j]);
      early2 = synthesize_augmented_early(SECONDARY, in2, c2);
      if (seg_n && !waited_w) HIPCALL(cq, vdf_ctx_sync_mark(cq, MARK_W));
      if (t_ahead) HIPCALL(ct, vdf_ctx_sync_mark(ct, MARK_T));
      HIPCALL(ctx, vdf_ctx_sync(ctx));
      const Field& Fb = *S1.Fb;
      // partial commitments leave the device as Jacobian points: summed as they are, one inversion for W and T together
      auto sum = [&](Pt acc, int from, int to) { for (int g = from; g < to; ++g) acc = pt_add(acc, pt_from_jac(hb[g], Fb), Fb); return acc; };
      const Pt w_sum = seg_n ? sum(pt_from_jac(p->h_pts[slot], Fb), 0, first_T) : pt_from_jac(hb[0], Fb);
      if (first) l1.comm_W = pt_to_aff(w_sum, Fb);
      else pt_to_aff2(w_sum, t_ahead ? sum(pt_add(pt_add(pt_from_jac(hb[4], Fb), pt_from_jac(hb[5], Fb), Fb), pt_from_jac(hb[6], Fb), Fb), first_T, ng)
                                     : pt_from_jac(hb[first_T], Fb), Fb, &l1.comm_W, &comm_T1);
      if (first) {
        // running primary := the fresh instance, relaxed; its A z, B z, C z once, folded from then on
        HIPCALL(ctx, vdf_dev_memcpy(ctx, s1.d_z, d_z2, S1.ncols * 32));
        HIPCALL(ctx, vdf_spmv3(ctx, S1.shape, (const vdf_fe*)s1.d_z, (vdf_fe*)s1.d_abc[0], (vdf_fe*)s1.d_abc[1], (vdf_fe*)s1.d_abc[2]));
      }
    }
    t4 = now_ms();
    if (!looked) {                                  // (the base step, a custom circuit, or the switch above)
      int rc = lookahead_advance();
      if (rc != VDF_OK) return rc;
    }
    return VDF_OK;
  }
  // ---- (d) the secondary augmented circuit, the primary fold, the next step's early rows set up ---------------------
  int chain_secondary_circuit() {
    {
      AugInputs& in = in2;
      in.u_W = l1.comm_W;
      in.T = comm_T1;
      CS cs(S2.field, false, pp->ro);
      Fe unew[9];
      const bool ahead_rows = pp->tune.nifs_ahead != 0;
      const bool rows_next = ahead_rows && !first && !custom && pp->ahead_rows != 0 && pp->ahead_mode != 1 && !p->ahead.empty();
      // The fold of the primary side and, behind it, the early rows of the NEXT step's cross term need nothing of this circuit
      // but its fold challenge r1 -- and from the fold to their commitment those rows are a step's longest dependent path
      // (fold, rows, sort, bucket accumulation, bucket reduction: ~0.7 ms; the chain waits for MARK_T half a step later).  The
      // synthesis derives r1 a few tens of microseconds into its late half and calls back (AugInputs::on_challenge): fold and
      // rows go out from THERE, under the rest of the synthesis (~0.1 ms of host work the device used to idle through on that
      // path), instead of after it.  `fold_and_rows(.., false)` after the synthesis is the same launches for the cases without
      // a call-back (base step, sequential synthesis, tuning.rows_at_challenge = 0).
      Fe u_folded = p->r[PRIMARY].inst.u;
      bool folded = false, rows_launched = false;
      auto fold_and_rows = [&](const uint64_t rch[4], bool launch_rows_now) -> int {
        folded = true;
        if (rows_next && p->ahead[0].slot != zin_slot) {
          // (not the slot the primary phase wrote z_in to: cannot happen with a lookahead of one step; copied again if it does)
          memcpy(p->h_zin + arity, p->zi[PRIMARY].data(), arity * 32);   // (this thread has waited for the primary side's launches: the slot's last copy is over)
          HIPCALL(ctx, vdf_dev_memcpy(ctx, (char*)p->d_z2s[p->ahead[0].slot] + (seg_b - arity) * 32, p->h_zin + arity, arity * 32));
          HIPCALL(ctx, vdf_ctx_mark(ctx, MARK_ZIN));
        }
        // The fold of the primary side goes to the queue of the early rows when they follow (they are what waits for it: the
        // rows then start right behind the fold's kernel, with no event between two queues, and the main queue goes straight to
        // the secondary side's NIFS); the main queue is made to wait for it at the end of the step, before anything reads the
        // folded instance there.  Everything the fold reads is complete: this thread has waited for the primary side's launches.
        // FUSED (tuning.fold_fused, the stencil in use): the early rows of A z, B z, C z and E are folded BY the next step's
        // stencil kernel on its way (it reads the previous fresh rows it is about to overwrite: one pass, no fold in front of it on
        // the step's longest dependent path); what is left -- z, and the ~10^4 rows outside the stencil -- is folded here on the
        // chain's queue, beside it: the two touch disjoint elements.
        const bool fused = rows_next && !first && stencil_fold_ok();
        vdf_ctx* fq = fused ? ctx : ((rows_next && fold_on_rows) ? ct : ctx);
        Fe rr = zero();
        if (!first) {
          SideState& s1 = p->r[PRIMARY];
          rr = int_to_fe(rch, F1);
          if (fused) {
            auto cut = [&](void* base, size_t b, size_t n) { return (vdf_fe*)((char*)base + b * 32); };
            vdf_fe* acc[8]; const vdf_fe* addv[8]; size_t len[8];
            int ns = 0;
            auto seg = [&](void* a, const void* b_, size_t begin, size_t n) { if (n) { acc[ns] = cut(a, begin, n); addv[ns] = cut((void*)b_, begin, n); len[ns] = n; ++ns; } };
            seg(s1.d_z, d_z2, 0, S1.ncols);
            seg(s1.d_E, s1.d_T, 0, ta_b); seg(s1.d_E, s1.d_T, ta_e, S1.num_cons - ta_e);
            HIPCALL(ctx, vdf_fold_many(ctx, S1.field, (const vdf_fe*)&rr, ns, acc, addv, (const size_t*)len));
            ns = 0;
            for (int m = 0; m < 3; ++m) { seg(s1.d_abc[m], s1.d_abc2[m], 0, ta_b); seg(s1.d_abc[m], s1.d_abc2[m], ta_e, S1.num_cons - ta_e); }
            if (ns) HIPCALL(ctx, vdf_fold_many(ctx, S1.field, (const vdf_fe*)&rr, ns, acc, addv, (const size_t*)len));
            p->poisoned = true;                                            // until the stencil that finishes this fold is on its queue
          } else {
            vdf_fe* acc[5] = {(vdf_fe*)s1.d_z, (vdf_fe*)s1.d_E, (vdf_fe*)s1.d_abc[0], (vdf_fe*)s1.d_abc[1], (vdf_fe*)s1.d_abc[2]};
            const vdf_fe* addv[5] = {(const vdf_fe*)d_z2, (const vdf_fe*)s1.d_T, (const vdf_fe*)s1.d_abc2[0], (const vdf_fe*)s1.d_abc2[1],
                                     (const vdf_fe*)s1.d_abc2[2]};
            const size_t len[5] = {S1.ncols, S1.num_cons, S1.num_cons, S1.num_cons, S1.num_cons};
            if (fq != ctx) HIPCALL(fq, vdf_ctx_wait_mark(fq, ctx, MARK_ZIN));         // (reached long ago; and the witness uploads in front of it)
            HIPCALL(fq, vdf_fold_many(fq, S1.field, (const vdf_fe*)&rr, 5, acc, addv, len));
          }
          u_folded = add(p->r[PRIMARY].inst.u, rr, F1);                   // u' = u + r: what the next step's rows are crossed with
        }
        // The early rows of the NEXT step's cross term: its rounds are in their ring slot (the lookahead), its input z_in is
        // this step's output (uploaded above, in front of the fold), the running instance is final once the fold is done: they
        // wait for MARK_FOLD, not for the uploads and the NIFS that follow.
        if (rows_next) {
          void* d_next = p->d_z2s[p->ahead[0].slot];
          vdf_ctx* cq_next = p->ctx2[(k + 1) % D];
          if (fused) HIPCALL(ct, vdf_ctx_wait_mark(ct, ctx, MARK_ZIN));    // z_in of the next step is in its slot (set in the primary phase)
          else HIPCALL(fq, vdf_ctx_mark(fq, MARK_FOLD));
          fold_elsewhere = fused || fq != ctx;
          if (launch_rows_now) {
            if (!fused && fq != ct) HIPCALL(ct, vdf_ctx_wait_mark(ct, fq, MARK_FOLD));
            p->r[PRIMARY].inst.u = u_folded;                             // (the launch reads it; the circuit's own value follows below)
            int rc = early_rows_launch(d_next, cq_next, true, fused ? &rr : nullptr);
            if (rc != VDF_OK) return rc;                                 // tahead_valid stays false: a retried step launches its rows itself
            p->poisoned = false;
            p->tahead_valid = true; p->tahead_slot = p->ahead[0].slot; p->tahead_k = k + 1; p->tahead_circuits = circuits;
            rows_launched = true;
          } else {
            deferred.pending = true; deferred.d_next = d_next; deferred.cq_next = cq_next; deferred.fq = fq;
            deferred.slot = p->ahead[0].slot;                            // (tahead_* are set once the rows are really on their queue)
            deferred.fused = fused; deferred.fold_r = rr; deferred.u_folded = u_folded;
          }
        }
        return VDF_OK;
      };
      int hook_rc = VDF_OK;
      if (!first && rows_next && pp->tune.rows_at_challenge)
        in.on_challenge = [&](const uint64_t rch[4]) { hook_rc = fold_and_rows(rch, true); };
      const std::vector<Fe> z_next = synthesize_augmented(cs, SECONDARY, in, c2, unew, r1, early2.get());
      in.on_challenge = nullptr;
      early2.reset();
      t5 = now_ms();
      if (hook_rc != VDF_OK) return hook_rc;
      if (!folded) { int rc = fold_and_rows(r1, false); if (rc != VDF_OK) return rc; }
      p->r[PRIMARY].inst = inst_from_elements(unew, F2, F1);           // base step: the first primary instance, relaxed
      // (the folded instance is in place: the launches below read its u)
      if (rows_launched && memcmp(&p->r[PRIMARY].inst.u, &u_folded, sizeof(Fe)) != 0)
        return fail(VDF_ERR_DEVICE, "prove_step: the circuit's folded u differs from the one the early rows were launched with");
      int rc = upload_fresh(ctx, S2, cs, p->h_stage[SECONDARY], p->d_l2z);
      if (rc != VDF_OK) return rc;
      p->l2.X[0] = cs.X[0]; p->l2.X[1] = cs.X[1];
      p->l2.u = one(F2);
      memset(&p->l2.comm_E, 0, sizeof(Aff));
      memset(&p->l2.comm_W, 0, sizeof(Aff));
      p->l2_committed = false;
      p->zi[SECONDARY] = z_next;
    }
    t6 = now_ms();
    return VDF_OK;
  }
  // ---- the way out: marks, and the next step's first device phases ---------------------------------------------------
  int chain_finish() {
    // the staging buffers are rewritten by the next call: their copies must have left (a mark), and nothing in flight may
    // read the circuits' memory once this call returns.  Behind that mark goes the next step's first device phase, which
    // needs nothing of the next step: the NIFS of the secondary instance just made (tuning.nifs_ahead = 0: left to the next call)
    const bool ahead = pp->tune.nifs_ahead != 0;
    HIPCALL(ctx, vdf_ctx_mark(ctx, MARK_STEP));
    if (ahead) {
      int rc = chain_launch_nifs2();
      if (rc != VDF_OK) return rc;
      p->nifs2 = vdf_proof::NIFS2_INFLIGHT;
    }
    if (deferred.pending) {                         // EARLY ROWS of the next step, behind the fold's mark
      if (!deferred.fused && deferred.fq != ct) HIPCALL(ct, vdf_ctx_wait_mark(ct, deferred.fq, MARK_FOLD));
      int rc = early_rows_launch(deferred.d_next, deferred.cq_next, true, deferred.fused ? &deferred.fold_r : nullptr);
      if (rc != VDF_OK) return rc;                  // tahead_valid stays false: a retried step launches its rows itself
      p->poisoned = false;
      p->tahead_valid = true; p->tahead_slot = deferred.slot; p->tahead_k = k + 1; p->tahead_circuits = circuits;
    }
    if (fold_elsewhere) HIPCALL(ctx, vdf_ctx_wait_mark(ctx, ct, MARK_FOLD));      // whatever reads the folded instance on the main queue comes after
    HIPCALL(ctx, vdf_ctx_sync_mark(ctx, MARK_STEP));
    for (int j = 0; j < D; ++j) if (touched[j]) HIPCALL(p->ctx2[j], vdf_ctx_sync_mark(p->ctx2[j], MARK_Z));
    p->i += 1;
    memcpy(&p->last.comm_W1, &l1.comm_W, sizeof(vdf_affine));
    memcpy(p->last.X1, l1.X, 64);
    memcpy(&p->last.comm_T1, &comm_T1, 64);
    memcpy(&p->last.comm_T2, &comm_T2, 64);
    memcpy(p->last.r1, r1, 32); memcpy(p->last.r2, r2, 32);
    return VDF_OK;
  }
};
}  // namespace

static int prove_step_impl(vdf_pp* pp, vdf_proof** proof, const vdf_circuits* circuits, size_t k, const vdf_step_circuit* custom,
                           const vdf_fe* z0, vdf_proof** fresh) {
  if (!pp || !proof || (!circuits && !custom) || !z0) return fail(VDF_ERR_BAD_ARG, "null argument");
  static const Circuit no_circuit{};
  if (!custom && k >= circuits->v.size()) return fail(VDF_ERR_BAD_LENGTH, "circuit index out of range");
  const Circuit& c = custom ? no_circuit : circuits->v[k];
  if (!custom && c.t != pp->t) return fail(VDF_ERR_BAD_LENGTH, "circuit t differs from the public parameters");
  const size_t arity = pp->arity;
  vdf_ctx* ctx = pp->ctx;
  vdf_proof* p = *proof;
  const bool first = (p == nullptr);
  if (first) {
    p = new vdf_proof();
    *fresh = p;
    p->pp = pp;
    p->z0[PRIMARY].assign((const Fe*)z0, (const Fe*)z0 + arity);
    p->z0[SECONDARY].assign(1, zero());                              // z0_secondary = [0], :310, :389-391
    p->zi[PRIMARY] = p->z0[PRIMARY];
    p->zi[SECONDARY] = p->z0[SECONDARY];
    int rc = alloc_proof_buffers(p);
    if (rc != VDF_OK) return rc;
  } else if (p->poisoned) {
    return fail(VDF_ERR_DEVICE, "this proof's running instance is half folded (an earlier prove_step failed on the device)");
  } else if (memcmp(p->z0[PRIMARY].data(), z0, 32 * arity) != 0) {
    return fail(VDF_ERR_BAD_ARG, "z0 differs from the one this proof was started with");
  }
  // StepCircuit::output's debug assertion: z_i must be the circuit's result (src/nova/proof.rs:147-149)
  if (!custom && memcmp(p->zi[PRIMARY].data(), &c.result, 96) != 0)
    return fail(VDF_ERR_BAD_ARG, "z_i does not match the circuit's result state");
  const double t0 = now_ms();
  int was_async = 0;
  HIPCALL(ctx, vdf_ctx_get_async(ctx, &was_async));
  HIPCALL(ctx, vdf_ctx_set_async(ctx, 1));
  struct Restore { vdf_ctx* c; int a; ~Restore() { if (!a) { vdf_ctx_sync(c); vdf_ctx_set_async(c, 0); } } } restore{ctx, was_async};
  StepRun run(pp, p, circuits, k, custom, c, first);
  run.t0 = run.t1 = run.t2 = run.t3 = run.t4 = run.t5 = run.t6 = t0;
  { int rc = run.lookahead_select(); if (rc != VDF_OK) return rc; }
  run.chain_prepare_inputs();
  { int rc = run.chain_secondary_nifs(); if (rc != VDF_OK) return rc; }
  { int rc = run.chain_primary_circuit(); if (rc != VDF_OK) return rc; }
  { int rc = run.chain_primary_nifs(); if (rc != VDF_OK) return rc; }
  { int rc = run.chain_secondary_circuit(); if (rc != VDF_OK) return rc; }
  { int rc = run.chain_finish(); if (rc != VDF_OK) return rc; }
  const double t7 = now_ms();
  p->ms[0] = run.t1 - t0; p->ms[1] = run.t2 - run.t1; p->ms[2] = run.t3 - run.t2; p->ms[3] = run.t4 - run.t3;
  p->ms[4] = run.t5 - run.t4; p->ms[5] = run.t6 - run.t5; p->ms[6] = t7 - run.t6; p->ms[7] = t7 - t0;
  *proof = p;
  *fresh = nullptr;
  return VDF_OK;
}

int vdf_nova_prove_recursively(vdf_pp* pp, const vdf_circuits* circuits, uint64_t num_iters_per_step, const vdf_fe z0[3],
                               vdf_proof** out) {
  return nova_guard([&]() -> int {
    if (!pp || !circuits || !out || !z0) return fail(VDF_ERR_BAD_ARG, "null argument");
    if (num_iters_per_step != pp->t) return fail(VDF_ERR_BAD_LENGTH, "num_iters_per_step differs from the public parameters");
    if (circuits->v.empty()) return fail(VDF_ERR_BAD_LENGTH, "no circuits (recursive_snark.unwrap(), src/nova/proof.rs:357)");
    vdf_proof* p = nullptr;
    for (size_t k = 0; k < circuits->v.size(); ++k) {                   // :318-355
      int rc = vdf_nova_prove_step(pp, &p, circuits, k, z0);
      if (rc != VDF_OK) { vdf_nova_proof_free(p); *out = nullptr; return rc; }
    }
    int rc = finalize_l2(p);
    if (rc != VDF_OK) { vdf_nova_proof_free(p); *out = nullptr; return rc; }
    *out = p;
    return VDF_OK;
  });
}

void vdf_nova_proof_free(vdf_proof* p) {
  if (!p) return;
  vdf_ctx* ctx = p->pp ? p->pp->ctx : nullptr;
  if (ctx) {
    for (vdf_ctx* q : p->ctx2) if (q) vdf_ctx_sync(q);                   // lookaheads may still be in flight
    if (p->ctx3) vdf_ctx_sync(p->ctx3);
    vdf_ctx_sync(ctx);
    for (SideState& st : p->r) {
      void* bufs[] = {st.d_z, st.d_E, st.d_T, st.d_abc[0], st.d_abc[1], st.d_abc[2], st.d_abc2[0], st.d_abc2[1], st.d_abc2[2]};
      for (void* b : bufs) if (b) vdf_dev_free(ctx, b);
    }
    if (p->d_l2z) vdf_dev_free(ctx, p->d_l2z);
    for (void* b : p->d_z2s) if (b) vdf_dev_free(ctx, b);
    for (void* b : p->d_traces) if (b) vdf_dev_free(ctx, b);
    for (void* b : p->d_packed) if (b) vdf_dev_free(ctx, b);
    for (vdf_ctx* q : p->ctx2) if (q) vdf_ctx_destroy(q);
    if (p->ctx3) vdf_ctx_destroy(p->ctx3);
    if (p->h_pts) vdf_host_free(ctx, p->h_pts);
    if (p->h_zin) vdf_host_free(ctx, p->h_zin);
    for (Fe* h : p->h_stage) if (h) vdf_host_free(ctx, h);
  }
  delete p;
}
size_t vdf_nova_proof_num_steps(const vdf_proof* p) { return p ? p->i : 0; }

int vdf_nova_proof_instance(const vdf_proof* p, int which, vdf_affine* comm_W, vdf_affine* comm_E, vdf_fe* u, vdf_fe X[2]) {
  return nova_guard([&]() -> int {
    if (!p || which < 0 || which > 2) return fail(VDF_ERR_BAD_ARG, "bad argument");
    if (which == VDF_INST_FRESH_SECONDARY) { int rc = finalize_l2(p); if (rc != VDF_OK) return rc; }
    const Inst& in = which == VDF_INST_FRESH_SECONDARY ? p->l2 : p->r[which].inst;
    if (comm_W) memcpy(comm_W, &in.comm_W, 64);
    if (comm_E) memcpy(comm_E, &in.comm_E, 64);
    if (u) memcpy(u, &in.u, 32);
    if (X) memcpy(X, in.X, 32 * NUM_IO);
    return VDF_OK;
  });
}
int vdf_nova_proof_witness_ptrs(const vdf_proof* p, int which, const void** d_z, const void** d_E) {
  return nova_guard([&]() -> int {
    if (!p || which < 0 || which > 3) return fail(VDF_ERR_BAD_ARG, "bad argument");
    if (p->poisoned) return fail(VDF_ERR_DEVICE, "this proof's running instance is half folded (an earlier prove_step failed on the device)");
    HIPCALL(p->pp->ctx, vdf_ctx_sync(p->pp->ctx));      // a step may have returned with its fold still in flight
    if (which == VDF_INST_FRESH_PRIMARY_LAST) { if (d_z) *d_z = p->d_z2s[p->slot]; if (d_E) *d_E = nullptr; }
    else if (which == VDF_INST_FRESH_SECONDARY) { if (d_z) *d_z = p->d_l2z; if (d_E) *d_E = nullptr; }
    else { if (d_z) *d_z = p->r[which].d_z; if (d_E) *d_E = p->r[which].d_E; }
    return VDF_OK;
  });
}
int vdf_nova_proof_zi(const vdf_proof* p, vdf_fe zi_primary[3], vdf_fe zi_secondary[1]) {
  if (!p) return fail(VDF_ERR_BAD_ARG, "null proof");
  if (zi_primary) memcpy(zi_primary, p->zi[PRIMARY].data(), 32 * p->pp->arity);
  if (zi_secondary) memcpy(zi_secondary, p->zi[SECONDARY].data(), 32);
  return VDF_OK;
}
int vdf_nova_proof_last_step(const vdf_proof* p, vdf_nova_step_info* out) {
  if (!p || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
  *out = p->last;
  return VDF_OK;
}
int vdf_nova_last_step_ms(const vdf_proof* p, double ms[8]) {
  if (!p || !ms) return fail(VDF_ERR_BAD_ARG, "null argument");
  memcpy(ms, p->ms, sizeof(p->ms));
  return VDF_OK;
}

// Per-launch timing of a prover's three queues (its context, the lookahead's, the early rows'): the three contexts'
// records merged on the device's common time line, queue = 0 / 1 / 2.
int vdf_nova_proof_set_kernel_timing(vdf_proof* p, int enable) {
  if (!p || !p->pp) return fail(VDF_ERR_BAD_ARG, "null argument");
  vdf_ctx* cs[3] = {p->pp->ctx, p->ctx2[0], p->ctx3};
  for (vdf_ctx* c : cs)
    if (c) HIPCALL(c, vdf_ctx_set_kernel_timing(c, enable));
  return VDF_OK;
}
int vdf_nova_proof_kernel_events(vdf_proof* p, vdf_kernel_event* out, int* queue, size_t cap, size_t* n) {
  if (!p || !p->pp || !n) return fail(VDF_ERR_BAD_ARG, "null argument");
  vdf_ctx* cs[3] = {p->pp->ctx, p->ctx2[0], p->ctx3};
  size_t cnt[3] = {0, 0, 0}, total = 0;
  for (int q = 0; q < 3; ++q)
    if (cs[q]) { HIPCALL(cs[q], vdf_ctx_kernel_events(cs[q], nullptr, 0, &cnt[q])); total += cnt[q]; }
  *n = total;
  if (!out) return VDF_OK;
  if (cap < total) return fail(VDF_ERR_BAD_LENGTH, "more launches recorded than the buffer holds");
  size_t at = 0;
  for (int q = 0; q < 3; ++q) {
    if (!cs[q]) continue;
    size_t got = 0;
    HIPCALL(cs[q], vdf_ctx_kernel_events(cs[q], out + at, cap - at, &got));
    if (queue) for (size_t i = 0; i < got; ++i) queue[at + i] = q;
    at += got;
  }
  *n = at;
  return VDF_OK;
}

// ---- verify --------------------------------------------------------------------------------------------
// RecursiveSNARK::verify(pp, num_steps, z0_primary, z0_secondary) -> (zi_primary, zi_secondary), then the comparison of
// src/nova/proof.rs:386 with z0_secondary = [0] (:389-391)
int vdf_nova_verify(const vdf_proof* p, vdf_pp* pp, size_t num_steps, const vdf_fe z0[3], const vdf_fe zi[3], int* ok) {
  return nova_guard([&]() -> int {
    return vdf_nova_verify_custom(p, pp, num_steps, z0, zi, ok);
  });
}

int vdf_nova_verify_custom(const vdf_proof* p, vdf_pp* pp, size_t num_steps, const vdf_fe* z0, const vdf_fe* zi, int* ok) {
  return nova_guard([&]() -> int {
    if (!p || !pp || !z0 || !zi || !ok) return fail(VDF_ERR_BAD_ARG, "null argument");
    *ok = 0;
    if (p->pp != pp) return fail(VDF_ERR_BAD_ARG, "proof was made under other public parameters");
    if (num_steps == 0 || p->i != num_steps) return VDF_OK;                  // NovaError::ProofVerifyError
    if (memcmp(p->z0[PRIMARY].data(), z0, 32 * pp->arity) != 0) return VDF_OK;   // not the chain this proof was started for
    { int rc = finalize_l2(p); if (rc != VDF_OK) return rc; }
    const Side& S1 = pp->s[PRIMARY];
    const Side& S2 = pp->s[SECONDARY];
    const Field& F1 = *S1.F;
    const Field& F2 = *S2.F;
    // (1) the two output hashes the last secondary instance carries
    const std::vector<Fe> z0p((const Fe*)z0, (const Fe*)z0 + pp->arity), z0s(1, zero());
    uint64_t hv[4];
    hash_state(S1.field, pp->params[PRIMARY], from_u64(num_steps, F1), z0p, p->zi[PRIMARY], to_relaxed(p->r[SECONDARY].inst, F2), hv, pp->ro);
    if (int_to_fe(hv, F2) != p->l2.X[0]) return VDF_OK;
    hash_state(S2.field, pp->params[SECONDARY], from_u64(num_steps, F2), z0s, p->zi[SECONDARY], to_relaxed(p->r[PRIMARY].inst, F1), hv, pp->ro);
    if (int_to_fe(hv, F2) != p->l2.X[1]) return VDF_OK;
    // (2) three satisfiability claims
    vdf_ctx* ctx = pp->ctx;
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    vdf_proof* q = const_cast<vdf_proof*>(p);                                  // scratch buffers only
    q->nifs2 = vdf_proof::NIFS2_NONE;          // ... which hold the next step's cross terms: that step makes them again
    if (q->ctx3) HIPCALL(q->ctx3, vdf_ctx_sync(q->ctx3));
    q->tahead_valid = false;
    bool good = false;
    for (int s = 0; s < 2; ++s) {
      int rc = check_sat(pp->s[s], p->r[s].inst, p->r[s].d_z, p->r[s].d_E, q->r[s].d_abc2, q->r[s].d_T, &good);
      if (rc != VDF_OK) return rc;
      if (!good) return VDF_OK;
    }
    if (p->l2.u != one(F2)) return VDF_OK;
    {
      int rc = check_sat(S2, p->l2, p->d_l2z, nullptr, q->r[SECONDARY].d_abc2, q->r[SECONDARY].d_T, &good);
      if (rc != VDF_OK) return rc;
      if (!good) return VDF_OK;
    }
    // Ok(zi_primary == zi && zi_secondary == [0]), src/nova/proof.rs:386
    *ok = (memcmp(p->zi[PRIMARY].data(), zi, 32 * pp->arity) == 0 && p->zi[SECONDARY][0].is_zero()) ? 1 : 0;
    return VDF_OK;
  });
}

}  // extern "C"
