// libvdf_nova.so, part 4: wire formats -- the step chain shared by both proof encodings, and the running proof
// (NovaVDFProof::Recursive) as a checkpoint a later process resumes from.  The reference keeps proofs in memory
// only (src/nova/proof.rs:52-55 derives no serialisation); the formats are this library's own.
#include "nova_internal.hpp"

using namespace vdfnova;

namespace vdfnova {

// magic[8] | t u64 | n u64 | digest[32] | z_0 [96] | per step k: z_{k+1} [96], comm_w_k [32], (k >= 1) comm_T_k [32]
// Field elements are canonical little-endian; points are the 32-byte encoding of host_math.hpp.  The challenges
// and the folded instance are not stored: the reader replays the folds, exactly as a verifier does.
size_t wire_chain_size(size_t n) { return 8 + 8 + 8 + 32 + 96 + n * (96 + 32) + (n ? n - 1 : 0) * 32; }

uint8_t* wire_put_chain(uint8_t* o, const char magic[8], uint64_t t, const uint8_t digest[32], const std::vector<StepRecord>& steps) {
  const Field& F = field(PRIMARY_FIELD);
  const Field& Fb = field_fp();
  const uint64_t n = steps.size();
  memcpy(o, magic, 8); o += 8;
  memcpy(o, &t, 8); o += 8;
  memcpy(o, &n, 8); o += 8;
  memcpy(o, digest, 32); o += 32;
  for (int j = 0; j < 3; ++j) o = wire_put_fe(o, steps[0].X[j], F);
  for (size_t k = 0; k < n; ++k) {
    for (int j = 3; j < 6; ++j) o = wire_put_fe(o, steps[k].X[j], F);
    pt_compress(steps[k].comm_w, Fb, o); o += 32;
    if (k) { pt_compress(steps[k].comm_T, Fb, o); o += 32; }
  }
  return o;
}

int wire_get_chain(const uint8_t** in, size_t* len, const char magic[8], const vdf_pp* pp, std::vector<StepRecord>* steps,
                   Aff* cW, Aff* cE, Fe* u, Fe X[NUM_IO]) {
  const Field& F = field(PRIMARY_FIELD);
  const Field& Fb = field_fp();
  const uint8_t* i = *in;
  if (*len < wire_chain_size(1)) return fail(VDF_ERR_BAD_LENGTH, "encoding is shorter than a one-step chain");
  if (memcmp(i, magic, 8) != 0) return fail(VDF_ERR_BAD_ARG, "not this kind of encoding (magic)");
  uint64_t t, n;
  memcpy(&t, i + 8, 8);
  memcpy(&n, i + 16, 8);
  if (t != pp->t || memcmp(i + 24, pp->digest, 32) != 0)
    return fail(VDF_ERR_BAD_ARG, "encoding was made under other public parameters");
  if (n == 0 || n > *len / 128 || wire_chain_size(n) > *len) return fail(VDF_ERR_BAD_LENGTH, "step count does not fit the encoding");
  i += 56;
  steps->assign(n, StepRecord());
  bool canonical = true, on_curve = true;
  Fe z[3];
  for (int j = 0; j < 3; ++j, i += 32) canonical &= wire_get_fe(i, F, &z[j]);
  for (size_t k = 0; k < n; ++k) {
    StepRecord& s = (*steps)[k];
    for (int j = 0; j < 3; ++j) s.X[j] = z[j];
    for (int j = 0; j < 3; ++j, i += 32) canonical &= wire_get_fe(i, F, &z[j]);
    for (int j = 0; j < 3; ++j) s.X[3 + j] = z[j];
    on_curve &= pt_decompress(i, Fb, &s.comm_w); i += 32;
    if (k) { on_curve &= pt_decompress(i, Fb, &s.comm_T); i += 32; }
    else s.comm_T.x = s.comm_T.y = zero();
  }
  if (!canonical) return fail(VDF_ERR_NONCANONICAL, "a field element of the chain is not canonical");
  if (!on_curve) return fail(VDF_ERR_NONCANONICAL, "a commitment of the chain does not decode to a curve point");
  std::vector<Fe> r(n);
  fold_replay(pp, *steps, r.data(), cW, cE, u, X);
  for (size_t k = 0; k < n; ++k) (*steps)[k].r = r[k];
  *len -= (size_t)(i - *in);
  *in = i;
  return VDF_OK;
}

}  // namespace vdfnova

extern "C" {

int vdf_nova_point_compress(const vdf_affine* p, uint8_t out[32]) {
  if (!p || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
  Aff a;
  memcpy(&a, p, sizeof(Aff));
  pt_compress(a, field_fp(), out);
  return VDF_OK;
}

int vdf_nova_point_decompress(const uint8_t in[32], vdf_affine* out) {
  if (!in || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
  Aff a;
  if (!pt_decompress(in, field_fp(), &a)) return fail(VDF_ERR_NONCANONICAL, "bytes decode to no point of the curve");
  memcpy(out, &a, sizeof(Aff));
  return VDF_OK;
}

// chain | W [num_vars x 32] | E [num_cons x 32]
size_t vdf_nova_proof_serialized_size(const vdf_proof* p) {
  if (!p || p->steps.empty()) return 0;
  return wire_chain_size(p->steps.size()) + 32 * (p->pp->num_vars + p->pp->num_cons);
}

int vdf_nova_proof_serialize(const vdf_proof* p, uint8_t* out, size_t cap) {
  if (!p || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
  if (p->steps.empty()) return fail(VDF_ERR_BAD_LENGTH, "nothing to serialise");
  if (cap < vdf_nova_proof_serialized_size(p)) return fail(VDF_ERR_BAD_LENGTH, "buffer too small");
  const vdf_pp* pp = p->pp;
  vdf_ctx* ctx = pp->ctx;
  const Field& F = field(PRIMARY_FIELD);
  uint8_t* o = wire_put_chain(out, WIRE_MAGIC_PROOF, pp->t, pp->digest, p->steps);
  // the witness leaves the device in Montgomery form (stream-ordered copy: every enqueued fold has landed)
  HIPCALL(ctx, vdf_dev_memcpy(ctx, o, p->d_z1, pp->num_vars * 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, o + pp->num_vars * 32, p->d_E, pp->num_cons * 32));
  for (size_t k = 0; k < pp->num_vars + pp->num_cons; ++k, o += 32) {
    Fe v;
    memcpy(v.l, o, 32);
    v = from_mont(v, F);
    memcpy(o, v.l, 32);
  }
  return VDF_OK;
}

int vdf_nova_proof_deserialize(vdf_pp* pp, const uint8_t* in, size_t len, vdf_proof** out) {
  if (!pp || !in || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
  *out = nullptr;
  vdf_ctx* ctx = pp->ctx;
  const Field& F = field(PRIMARY_FIELD);
  struct Guard { vdf_proof* p; ~Guard() { if (p) vdf_nova_proof_free(p); } } g{new vdf_proof()};
  vdf_proof* p = g.p;
  p->pp = pp;
  int rc = wire_get_chain(&in, &len, WIRE_MAGIC_PROOF, pp, &p->steps, &p->comm_W, &p->comm_E, &p->u, p->X);
  if (rc != VDF_OK) return rc;
  const size_t nv = pp->num_vars, nc = pp->num_cons;
  if (len != 32 * (nv + nc)) return fail(VDF_ERR_BAD_LENGTH, "witness section has the wrong length for this shape");
  p->i = p->steps.size();
  for (int j = 0; j < 3; ++j) p->zi[j] = p->steps.back().X[3 + j];
  std::vector<Fe> z(pp->ncols), E(nc);
  bool canonical = true;
  for (size_t k = 0; k < nv; ++k) canonical &= wire_get_fe(in + 32 * k, F, &z[k]);
  for (size_t k = 0; k < nc; ++k) canonical &= wire_get_fe(in + 32 * (nv + k), F, &E[k]);
  if (!canonical) return fail(VDF_ERR_NONCANONICAL, "a field element of the witness is not canonical");
  for (size_t k = nv; k < pp->ncols; ++k) z[k] = zero();
  z[nv] = p->u;
  for (int j = 0; j < NUM_IO; ++j) z[nv + 1 + j] = p->X[j];
  rc = alloc_proof_buffers(p);
  if (rc != VDF_OK) return rc;
  HIPCALL(ctx, vdf_dev_memcpy(ctx, p->d_z1, z.data(), pp->ncols * 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, p->d_E, E.data(), nc * 32));
  // A z, B z, C z of the running instance are state the prover folds instead of recomputing: rebuild them
  HIPCALL(ctx, vdf_spmv3(ctx, pp->shape, (const vdf_fe*)p->d_z1, (vdf_fe*)p->d_abc[0], (vdf_fe*)p->d_abc[1], (vdf_fe*)p->d_abc[2]));
  // a checkpoint whose witness does not open its own folded commitments would only fail much later, at verify
  vdf_jac jw, je;
  HIPCALL(ctx, vdf_msm(ctx, pp->gens, 0, (const vdf_fe*)p->d_z1, nv, 1, &jw));
  HIPCALL(ctx, vdf_msm(ctx, pp->gens, 0, (const vdf_fe*)p->d_E, nc, 1, &je));
  HIPCALL(ctx, vdf_ctx_sync(ctx));
  Aff aw, ae;
  aw = jac_to_aff(jw, field_fp());
  ae = jac_to_aff(je, field_fp());
  if (memcmp(&aw, &p->comm_W, sizeof(Aff)) || memcmp(&ae, &p->comm_E, sizeof(Aff)))
    return fail(VDF_ERR_BAD_ARG, "the witness does not open the commitments its step records fold to");
  *out = p;
  g.p = nullptr;
  return VDF_OK;
}

}  // extern "C"
