/* vdf_nova.h -- host layer above the kernel ABI (include/vdf_hip.h): the reference crate's public
 * surface for the MinRoot VDF and its Nova proof, restated as a C ABI (libvdf_nova.so).
 *
 * The reference is Rust (/root/reference/src/minroot.rs, src/nova/proof.rs); no Rust toolchain
 * exists in the build image, so the host side is C++ and mirrors the reference's names, argument
 * meaning and error behaviour.  Every heavy operation is a call through include/vdf_hip.h; this
 * library contains no kernel and no CPU fallback for one.  What stays on the host is what the
 * reference also keeps sequential or O(1): the forward MinRoot evaluation (src/minroot.rs:329-359,
 * the "delay" itself), transcript hashing, and the two scalar multiplications of an instance fold.
 *
 * STAGE: folding-only, plus the compression SNARK over the folded instance.  A step is the reference's step circuit
 * (InverseMinRootCircuit::synthesize, src/nova/proof.rs:87-140) wrapped so that z_in / z_out are
 * public (the "exposed-IO wrapper", oracle/pasta.py step_circuit_shape) and folded with NIFS
 * (SURVEY.md Appendix C).  The in-circuit verifier (augmented circuit), the secondary curve and
 * Poseidon are not built: the proof is therefore linear in the number of steps (the verifier
 * replays the folds) instead of constant-size.  The transcript hash is SHAKE256, squeezed to 128
 * bits; nova-snark's Poseidon transcript is implementation-defined and unpinned.
 */
#ifndef VDF_NOVA_H
#define VDF_NOVA_H

#include "vdf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* src/minroot.rs:14-20 */
enum { VDF_MODE_LTR_SEQUENTIAL = 0, VDF_MODE_LTR_ADDCHAIN_SEQUENTIAL = 1, VDF_MODE_RTL_SEQUENTIAL = 2,
       VDF_MODE_RTL_ADDCHAIN_SEQUENTIAL = 3 };

/* src/minroot.rs:267-272: State<T> { x, y, i } as three Montgomery field elements */
typedef struct { vdf_fe x, y, i; } vdf_state;

/* ---- MinRoot evaluator (trait MinRootVDF, src/minroot.rs:287-374); field = VDF_FIELD_FQ is
 * PallasVDF (:40-197), VDF_FIELD_FP is VestaVDF (:201-262, ignores the mode, :203-205) -------- */
int vdf_minroot_forward_step(int field, int mode, const vdf_fe* x, vdf_fe* out);      /* :77-84, :223 */
int vdf_minroot_inverse_step(int field, const vdf_fe* x, vdf_fe* out);                /* :73-75 */
int vdf_minroot_round(int field, int mode, const vdf_state* s, vdf_state* out);       /* :329-335 */
int vdf_minroot_inverse_round(int field, const vdf_state* s, vdf_state* out);         /* :338-344 */
/* eval / simple_eval (:348-359).  trace_xy (optional, 2*(t+1) elements) receives (x, y) of every
 * state 0..t -- the reference discards it (:352-359); the witness kernel consumes it. */
int vdf_minroot_eval(int field, int mode, const vdf_state* s, uint64_t t, vdf_state* out, vdf_fe* trace_xy);
int vdf_minroot_inverse_eval(int field, const vdf_state* s, uint64_t t, vdf_state* out);   /* :363-365 */
int vdf_minroot_check(int field, const vdf_state* result, uint64_t t, const vdf_state* original); /* :369-371; 1 = ok */
int vdf_minroot_element(int field, uint64_t n, vdf_fe* out);                           /* V::element, :60, :207 */

/* ---- Nova proof API (src/nova/proof.rs:232-392) ------------------------------------------------ */
typedef struct vdf_pp vdf_pp;             /* NovaVDFPublicParams, :38-43 */
typedef struct vdf_circuits vdf_circuits; /* Vec<InverseMinRootCircuit<G1>>, :57-66 (reversed, :294) */
typedef struct vdf_proof vdf_proof;       /* NovaVDFProof::Recursive, :51-55 */
typedef struct vdf_snark vdf_snark;       /* NovaVDFProof::Compressed, :54 */

/* public_params(num_iters_per_step), :232-237: R1CS shape of the wrapped step circuit, Pedersen
 * generators (next_pow2(max(vars, cons)) of them by seeded try-and-increment -- unknown discrete logarithms,
 * VDF_GENS_TRY_AND_INCREMENT; nova-snark's own label -> hash derivation is unpinned) with their fixed-base table,
 * shape digest.  One-time; outside every timed region (benches/nova.rs:51-58). */
int  vdf_nova_public_params(vdf_ctx* ctx, uint64_t num_iters_per_step, vdf_pp** out);
void vdf_nova_pp_free(vdf_pp* pp);
int  vdf_nova_pp_sizes(const vdf_pp* pp, uint64_t* num_cons, uint64_t* num_vars, uint64_t* num_io, uint64_t* nnz3,
                       uint64_t* num_gens);

/* InverseMinRootCircuit::eval_and_make_circuits, :262-299: num_steps forward evaluations of
 * num_iters_per_step rounds each from initial_state (host, sequential), one circuit per step
 * holding (result, input) and -- beyond the reference -- the step's forward trace; circuits are
 * returned REVERSED (:294).  z0_primary = the final state (:278-281). */
int  vdf_nova_eval_and_make_circuits(int mode, uint64_t num_iters_per_step, size_t num_steps,
                                     const vdf_state* initial_state, vdf_fe z0_primary[3], vdf_circuits** out);
/* Moves every circuit's forward trace into HBM (outside the timed region: the trace is an input of
 * proving, produced by the untimed forward evaluation). */
int  vdf_nova_circuits_upload(vdf_ctx* ctx, vdf_circuits* c);
size_t vdf_nova_circuits_len(const vdf_circuits* c);
/* result / input of circuit k (k = 0 is proved first): InverseMinRootCircuit.result / .input */
int  vdf_nova_circuit_states(const vdf_circuits* c, size_t k, vdf_state* result, vdf_state* input);
void vdf_nova_circuits_free(vdf_circuits* c);

/* NovaVDFProof::prove_recursively(pp, circuits, num_iters_per_step, z0), :302-358: one prove_step
 * per circuit.  Returns VDF_ERR_* (the reference asserts success, :353). */
int  vdf_nova_prove_recursively(vdf_pp* pp, const vdf_circuits* circuits, uint64_t num_iters_per_step,
                                const vdf_fe z0[3], vdf_proof** out);
/* One RecursiveSNARK::prove_step (:342-349): *proof == NULL starts a new proof (the `None` case).
 * What a step computes that does not depend on the chain -- the fresh witness of circuit k and its commitment -- is
 * enqueued one call early, for circuit k + 1 of the same `circuits`, on a second context the proof owns; a call for any
 * other step simply finds no such work waiting and does it then.  The results are those of a prover without lookahead,
 * and no call returns while anything in flight still reads the circuits' memory (they may be freed right after).
 * The fresh commitment is an MSM over 3t + 4 merged generators (vdf_minroot_step_z_packed, include/vdf_hip.h): the
 * same point as the commitment to all 4t + 4 witness values. */
int  vdf_nova_prove_step(vdf_pp* pp, vdf_proof** proof, const vdf_circuits* circuits, size_t k, const vdf_fe z0[3]);
/* NovaVDFProof::verify(pp, num_steps, z0, zi), :370-387: *ok = 1 iff the proof is valid for
 * num_steps steps from z0 AND the verified zi_primary equals zi (the Ok(bool) of :386). */
int  vdf_nova_verify(const vdf_proof* proof, vdf_pp* pp, size_t num_steps, const vdf_fe z0[3], const vdf_fe zi[3], int* ok);
void vdf_nova_proof_free(vdf_proof* proof);
size_t vdf_nova_proof_num_steps(const vdf_proof* proof);
/* Introspection for parity tests: running relaxed instance (commitments as affine points, u, X[6])
 * and device pointers of the running witness W (num_vars) and error vector E (num_cons). */
int  vdf_nova_proof_instance(const vdf_proof* proof, vdf_affine* comm_W, vdf_affine* comm_E, vdf_fe* u, vdf_fe X[6]);
int  vdf_nova_proof_witness_ptrs(const vdf_proof* proof, const void** d_W, const void** d_E);
/* per-step record k: fresh commitment, cross-term commitment, challenge, public IO */
int  vdf_nova_proof_step_record(const vdf_proof* proof, size_t k, vdf_affine* comm_w, vdf_affine* comm_T, vdf_fe* r, vdf_fe X[6]);
/* host wall-clock of the last prove_step, milliseconds.  A step is enqueued asynchronously, so the slots are
 * launch times except [3] and [4]: fresh witness + its commitment (zero when the previous step looked ahead),
 * launch of the commitment of T, cross-term launch, wait for the fresh commitment + launch of the next step's
 * lookahead, wait (previous step's host instance fold, then the commitment of T), transcript + fold launch,
 * bookkeeping, total. */
int  vdf_nova_last_step_ms(const vdf_proof* proof, double ms[8]);

/* ---- compression (src/nova/proof.rs:360-368, :383) ---------------------------------------------------------
 * NovaVDFProof::compress -> nova-snark CompressedSNARK::prove: a succinct argument that the folded relaxed R1CS
 * instance is satisfiable, instead of its 14 MB witness.  Protocol "vdf-spartan-v3" (oracle/spartan.py): a
 * Spartan-style sum-check argument with inner-product-argument openings under the same Pedersen generators; the
 * extra generator of the openings is generator number num_gens of the same family.  Like the rest of this layer it is
 * self-consistent, not interchangeable with nova-snark (whose constants are unpinned); in the
 * folding-only stage the compressed proof still carries the per-step records the verifier replays.  Every pass
 * over a vector runs on the GPU through include/vdf_hip.h. */
int  vdf_nova_compress(const vdf_proof* proof, vdf_pp* pp, vdf_snark** out);
/* NovaVDFProof::verify for the Compressed variant: *ok = 1 iff the step records chain from z0 to zi over num_steps
 * steps, fold to the stated instance, and the argument for that instance verifies. */
int  vdf_nova_verify_compressed(const vdf_snark* snark, vdf_pp* pp, size_t num_steps, const vdf_fe z0[3], const vdf_fe zi[3],
                                int* ok);
void vdf_nova_snark_free(vdf_snark* snark);
/* Flat canonical encoding of the argument (little-endian, non-Montgomery; layout in the implementation and in
 * tests/test_gpu_compress.py): size, export, and import -- which replaces the argument of `snark` and returns
 * VDF_ERR_NONCANONICAL for out-of-range field elements. */
size_t vdf_nova_snark_size(const vdf_snark* snark);
int  vdf_nova_snark_bytes(const vdf_snark* snark, uint8_t* out, size_t cap);
int  vdf_nova_snark_set_bytes(vdf_snark* snark, const uint8_t* in, size_t len);

/* ---- wire formats ---------------------------------------------------------------------
 * The reference keeps its proofs in memory only (src/nova/proof.rs:52-55 derives no serialisation); these encodings
 * are this library's own, versioned by their magic.  Field elements: 32 bytes, canonical, little-endian.  Points:
 * 32 bytes, canonical little-endian x with the parity of y in bit 255, the identity as 32 zero bytes.
 *
 *   chain  = magic[8] | t u64 | num_steps u64 | digest of the public parameters [32] | z_0 [96]
 *            | per step k: z_{k+1} [96], commitment of the fresh witness [32], (k >= 1) cross-term commitment [32]
 *   "VDFSNK02" compressed proof = chain | the argument of vdf_nova_snark_bytes with 32-byte points
 *   "VDFRSK01" running proof    = chain | W [num_vars x 32] | E [num_cons x 32]
 *
 * Challenges and the folded instance are not stored: deserialisation replays the folds (as verification does), so a
 * decoded proof states nothing the reader did not derive.  Deserialisation fails with VDF_ERR_BAD_ARG for a foreign
 * magic or other public parameters, VDF_ERR_BAD_LENGTH for a length that does not fit the shape, and
 * VDF_ERR_NONCANONICAL for an out-of-range field element or bytes that decode to no curve point.
 *
 * The compressed proof is what a prover ships to a verifier in another process: 56 + 96 + 160 n - 32 bytes of chain
 * plus 5.9 KB of argument at t = 2^16.  The running proof is a checkpoint: vdf_nova_proof_deserialize rebuilds the
 * device-resident state (including A z, B z, C z of the running instance), refuses a witness that does not open the
 * commitments its records fold to, and vdf_nova_prove_step continues from it. */
/* The 32-byte point encoding by itself (host arithmetic only, no device): commitments are points of Pallas, in
 * Montgomery coordinates like everywhere in this ABI.  decompress: VDF_ERR_NONCANONICAL unless the bytes are exactly
 * what compress writes for some point. */
int  vdf_nova_point_compress(const vdf_affine* p, uint8_t out[32]);
int  vdf_nova_point_decompress(const uint8_t in[32], vdf_affine* out);
size_t vdf_nova_snark_serialized_size(const vdf_snark* snark);
int  vdf_nova_snark_serialize(const vdf_snark* snark, uint8_t* out, size_t cap);
int  vdf_nova_snark_deserialize(vdf_pp* pp, const uint8_t* in, size_t len, vdf_snark** out);
size_t vdf_nova_proof_serialized_size(const vdf_proof* proof);
int  vdf_nova_proof_serialize(const vdf_proof* proof, uint8_t* out, size_t cap);
int  vdf_nova_proof_deserialize(vdf_pp* pp, const uint8_t* in, size_t len, vdf_proof** out);
const char* vdf_nova_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* VDF_NOVA_H */
