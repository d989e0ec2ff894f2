// libvdf_nova.so, part 4: wire formats -- the 32-byte point encoding and the running proof (NovaVDFProof::Recursive)
// as a checkpoint a later process resumes from.  The reference keeps proofs in memory only (src/nova/proof.rs:52-55
// derives no serialisation); the formats are this library's own (layouts in include/vdf_nova.h,
// restated in oracle/wire.py).  The compressed proof's encoding lives next to the argument in compress_host.cpp.
#include "nova_internal.hpp"

using namespace vdfnova;

extern "C" {

int vdf_nova_point_compress(int curve, const vdf_affine* p, uint8_t out[32]) {
  if (!p || !out || (curve != VDF_CURVE_PALLAS && curve != VDF_CURVE_VESTA)) return fail(VDF_ERR_BAD_ARG, "bad argument");
  Aff a;
  memcpy(&a, p, sizeof(Aff));
  pt_compress(a, field(curve == VDF_CURVE_PALLAS ? VDF_FIELD_FP : VDF_FIELD_FQ), out);
  return VDF_OK;
}

int vdf_nova_point_decompress(int curve, const uint8_t in[32], vdf_affine* out) {
  if (!in || !out || (curve != VDF_CURVE_PALLAS && curve != VDF_CURVE_VESTA)) return fail(VDF_ERR_BAD_ARG, "bad argument");
  Aff a;
  if (!pt_decompress(in, field(curve == VDF_CURVE_PALLAS ? VDF_FIELD_FP : VDF_FIELD_FQ), &a))
    return fail(VDF_ERR_NONCANONICAL, "bytes decode to no point of the curve");
  memcpy(out, &a, sizeof(Aff));
  return VDF_OK;
}

// "VDFRSK02": magic[8] | t u64 | i u64 | digest[32] | z_0 [96] | z_i primary [96] | z_i secondary [32]
//   | running primary instance [160] | running secondary instance [160] | fresh secondary instance [96]
//   | W1 | E1 | W2 | E2 | w2   (num_vars / num_cons elements of the side, 32 bytes each, canonical)
static size_t header_size(const vdf_pp* pp) { return 8 + 8 + 8 + 32 + 64 * pp->arity + 32 + 2 * INST_WIRE_RELAXED + INST_WIRE_STRICT; }

size_t vdf_nova_proof_serialized_size(const vdf_proof* p) {
  if (!p || p->i == 0) return 0;
  const vdf_pp* pp = p->pp;
  return header_size(pp) + 32 * (pp->s[0].num_vars + pp->s[0].num_cons + 2 * pp->s[1].num_vars + pp->s[1].num_cons);
}

int vdf_nova_proof_serialize(const vdf_proof* p, uint8_t* out, size_t cap) {
  return nova_guard([&]() -> int {
    if (!p || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
    if (p->i == 0) return fail(VDF_ERR_BAD_LENGTH, "nothing to serialise");
    if (cap < vdf_nova_proof_serialized_size(p)) return fail(VDF_ERR_BAD_LENGTH, "buffer too small");
    { int rc = finalize_l2(p); if (rc != VDF_OK) return rc; }
    const vdf_pp* pp = p->pp;
    vdf_ctx* ctx = pp->ctx;
    const uint64_t steps = p->i;
    uint8_t* o = out;
    memcpy(o, WIRE_MAGIC_PROOF, 8); o += 8;
    memcpy(o, &pp->t, 8); o += 8;
    memcpy(o, &steps, 8); o += 8;
    memcpy(o, pp->digest, 32); o += 32;
    for (size_t k = 0; k < pp->arity; ++k) o = wire_put_fe(o, p->z0[PRIMARY][k], *pp->s[0].F);
    for (size_t k = 0; k < pp->arity; ++k) o = wire_put_fe(o, p->zi[PRIMARY][k], *pp->s[0].F);
    o = wire_put_fe(o, p->zi[SECONDARY][0], *pp->s[1].F);
    o = put_inst(o, p->r[0].inst, pp->s[0], true);
    o = put_inst(o, p->r[1].inst, pp->s[1], true);
    o = put_inst(o, p->l2, pp->s[1], false);
    // the witnesses leave the device in Montgomery form (stream-ordered copies: every enqueued fold has landed)
    struct Vec { const void* d; size_t n; const Field* F; };
    const Vec vecs[5] = {{p->r[0].d_z, pp->s[0].num_vars, pp->s[0].F}, {p->r[0].d_E, pp->s[0].num_cons, pp->s[0].F},
                         {p->r[1].d_z, pp->s[1].num_vars, pp->s[1].F}, {p->r[1].d_E, pp->s[1].num_cons, pp->s[1].F},
                         {p->d_l2z, pp->s[1].num_vars, pp->s[1].F}};
    for (const Vec& v : vecs) {
      HIPCALL(ctx, vdf_dev_memcpy(ctx, o, v.d, v.n * 32));
      for (size_t k = 0; k < v.n; ++k, o += 32) {
        Fe e;
        memcpy(e.l, o, 32);
        e = from_mont(e, *v.F);
        memcpy(o, e.l, 32);
      }
    }
    return VDF_OK;
  });
}

int vdf_nova_proof_deserialize(vdf_pp* pp, const uint8_t* in, size_t len, vdf_proof** out) {
  return nova_guard([&]() -> int {
    if (!pp || !in || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    vdf_ctx* ctx = pp->ctx;
    if (len < header_size(pp)) return fail(VDF_ERR_BAD_LENGTH, "encoding is shorter than its header");
    if (memcmp(in, WIRE_MAGIC_PROOF, 8) != 0) return fail(VDF_ERR_BAD_ARG, "not this kind of encoding (magic)");
    uint64_t t, steps;
    memcpy(&t, in + 8, 8);
    memcpy(&steps, in + 16, 8);
    if (t != pp->t || memcmp(in + 24, pp->digest, 32) != 0) return fail(VDF_ERR_BAD_ARG, "encoding was made under other public parameters");
    const Side& S1 = pp->s[0];
    const Side& S2 = pp->s[1];
    if (steps == 0 || len != header_size(pp) + 32 * (S1.num_vars + S1.num_cons + 2 * S2.num_vars + S2.num_cons))
      return fail(VDF_ERR_BAD_LENGTH, "encoding has the wrong length for this shape");
    struct Guard { vdf_proof* p; ~Guard() { if (p) vdf_nova_proof_free(p); } } g{new vdf_proof()};
    vdf_proof* p = g.p;
    p->pp = pp;
    p->i = steps;
    const uint8_t* i = in + 56;
    bool canonical = true, on_curve = true;
    p->z0[PRIMARY].resize(pp->arity); p->zi[PRIMARY].resize(pp->arity);
    p->z0[SECONDARY].assign(1, zero()); p->zi[SECONDARY].resize(1);
    for (size_t k = 0; k < pp->arity; ++k, i += 32) canonical &= wire_get_fe(i, *S1.F, &p->z0[PRIMARY][k]);
    for (size_t k = 0; k < pp->arity; ++k, i += 32) canonical &= wire_get_fe(i, *S1.F, &p->zi[PRIMARY][k]);
    canonical &= wire_get_fe(i, *S2.F, &p->zi[SECONDARY][0]); i += 32;
    Inst r1, r2, l2;
    i = get_inst(i, &r1, S1, true, &canonical, &on_curve);
    i = get_inst(i, &r2, S2, true, &canonical, &on_curve);
    i = get_inst(i, &l2, S2, false, &canonical, &on_curve);
    if (!on_curve) return fail(VDF_ERR_NONCANONICAL, "a commitment does not decode to a curve point");
    int rc = alloc_proof_buffers(p);
    if (rc != VDF_OK) return rc;
    p->r[0].inst = r1; p->r[1].inst = r2; p->l2 = l2;
    p->l2_committed = true;
    struct Vec { void* d; size_t n, total; const Side* sd; const Inst* inst; };
    const Vec vecs[5] = {{p->r[0].d_z, S1.num_vars, S1.ncols, &S1, &r1}, {p->r[0].d_E, S1.num_cons, S1.num_cons, &S1, nullptr},
                         {p->r[1].d_z, S2.num_vars, S2.ncols, &S2, &r2}, {p->r[1].d_E, S2.num_cons, S2.num_cons, &S2, nullptr},
                         {p->d_l2z, S2.num_vars, S2.ncols, &S2, &l2}};
    std::vector<Fe> buf;
    for (const Vec& v : vecs) {
      buf.assign(v.total, zero());
      for (size_t k = 0; k < v.n; ++k, i += 32) canonical &= wire_get_fe(i, *v.sd->F, &buf[k]);
      if (v.inst) {                                      // z = [W | u | X]
        buf[v.n] = v.inst->u;
        for (int j = 0; j < NUM_IO; ++j) buf[v.n + 1 + j] = v.inst->X[j];
      }
      HIPCALL(ctx, vdf_dev_memcpy(ctx, v.d, buf.data(), v.total * 32));
    }
    if (!canonical) return fail(VDF_ERR_NONCANONICAL, "a field element of the encoding is not canonical");
    // A z, B z, C z of the running instances are state the prover folds instead of recomputing: rebuild them
    for (int s = 0; s < 2; ++s)
      HIPCALL(ctx, vdf_spmv3(ctx, pp->s[s].shape, (const vdf_fe*)p->r[s].d_z, (vdf_fe*)p->r[s].d_abc[0], (vdf_fe*)p->r[s].d_abc[1],
                             (vdf_fe*)p->r[s].d_abc[2]));
    // a checkpoint whose witnesses do not open its commitments would only fail much later, at verify
    struct Open { const Side* sd; const void* d; size_t n; const Aff* want; };
    const Open opens[5] = {{&S1, p->r[0].d_z, S1.num_vars, &r1.comm_W}, {&S1, p->r[0].d_E, S1.num_cons, &r1.comm_E},
                           {&S2, p->r[1].d_z, S2.num_vars, &r2.comm_W}, {&S2, p->r[1].d_E, S2.num_cons, &r2.comm_E},
                           {&S2, p->d_l2z, S2.num_vars, &l2.comm_W}};
    for (const Open& op : opens) {
      vdf_jac j;
      HIPCALL(ctx, vdf_msm(ctx, op.sd->gens, 0, (const vdf_fe*)op.d, op.n, 1, &j));
      const Aff a = jac_to_aff(j, *op.sd->Fb);
      if (memcmp(&a, op.want, sizeof(Aff))) return fail(VDF_ERR_BAD_ARG, "a witness does not open its commitment");
    }
    *out = p;
    g.p = nullptr;
    return VDF_OK;
  });
}

}  // extern "C"
