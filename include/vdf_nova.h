/* vdf_nova.h -- host layer above the kernel ABI (include/vdf_hip.h): the reference crate's public
 * surface for the MinRoot VDF and its Nova proof, restated as a C ABI (libvdf_nova.so).
 *
 * The reference is Rust (/root/reference/src/minroot.rs, src/nova/proof.rs); no Rust toolchain
 * exists in the build image, so the host side is C++ and mirrors the reference's names, argument
 * meaning and error behaviour.  Every pass over a vector is a call through include/vdf_hip.h; this
 * library contains no kernel and no CPU fallback for one.  What stays on the host is what the
 * reference also keeps sequential or O(1): the forward MinRoot evaluation (src/minroot.rs:329-359,
 * the "delay" itself), and the synthesis of the ~10^4 variables of an augmented circuit that are not
 * MinRoot rounds (hashes, in-circuit group arithmetic: strictly sequential field work).
 *
 * What a proof is: Nova IVC on the Pallas / Vesta cycle (src/nova/proof.rs:26-43), protocol
 * "vdf-nova-ivc-v1", specified in oracle/nova.py.  prove_step (:342-349) folds the previous secondary
 * instance, synthesises the PRIMARY augmented circuit (in-circuit check of the previous output hash,
 * in-circuit NIFS verifier for the secondary curve, the MinRoot step circuit, the next output hash)
 * commits and folds it, then does the same for the SECONDARY augmented circuit around
 * TrivialTestCircuit (:258-260).  A proof has constant size in the number of steps and verify
 * (:370-392) recomputes two hashes and checks three satisfiability claims, `zi_secondary == [0]`
 * included.  nova-snark 0.8.0's own constants (Poseidon parameters, allocation order, transcript,
 * generator derivation) are not in /root/reference and pinned by none of its tests:
 * proofs are self-consistent and bit-exact against oracle/nova.py, not interchangeable with
 * nova-snark's -- parity unpinned.
 *
 * Threading.  The evaluator functions and vdf_nova_ro_hash / shape_digest / aug_synthesize are pure (any thread, any time).
 * A parameter set (vdf_pp) owns the device queue its proofs enqueue on: ONE call at a time per parameter set and per
 * handle made under it (prove_step, verify, compress, the wire functions that touch the device); calls under different
 * parameter sets are independent and may run on different threads at once (two chains = two sets: bench.py's
 * aggregate_over_concurrent_chains).  Concurrent vdf_nova_compress calls under one set are serialised by the library at the
 * point where they share its second queue; nothing else is.  The library has threads of its own: up to a few sets of three
 * detached synthesis helpers, made when a prover first wants them and parked between calls (they never touch a handle
 * outside the call that handed them work), and one thread per vdf_nova_compress for the secondary side's argument, joined
 * before the call returns.
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

/* The primary step circuit (the seam of :79-153).  REFERENCE (the default: what vdf_nova_public_params builds, what
 * the bench measures and the parity tests check first): the circuit exactly as the reference writes it -- 4 variables
 * per round, new_x allocated at :167-173 and used by no constraint (:219-227 takes y - i + 1 directly), 4t + 1 step
 * variables, 3t + 1 constraints; 2^19 generators at t = 2^16.  Because nothing binds new_x a prover may set it freely, so
 * a proof over this circuit does not attest the VDF (tests/test_oracle_nova.py builds the forgery): a weakness of the
 * work-in-progress reference that a drop-in reproduces.  BOUND: the sound variant, offered as an option -- new_x carried
 * as the linear combination y - i + 1 instead of a variable, 3 variables and the same 3 constraints per round.
 * The witness of the reference's circuit is committed WITHOUT a term per new_x: new_x_j is an affine image of new_y_(j-1),
 * so its share folds into derived generators (vdf_hip.h vdf_minroot_step_segment_packed) -- same commitment, 3t + 4 terms. */
enum { VDF_CIRCUIT_MINROOT_BOUND = 0, VDF_CIRCUIT_MINROOT_REFERENCE = 1, VDF_CIRCUIT_CUSTOM = 2 /* vdf_step_circuit, below */ };
enum { VDF_SIDE_PRIMARY = 0, VDF_SIDE_SECONDARY = 1 };

/* public_params(num_iters_per_step), :232-237: both augmented circuits synthesised once for their R1CS shapes,
 * Pedersen generators per curve (next_pow2(max(vars, cons)) of them, seeded try-and-increment -- unknown discrete
 * logarithms; nova-snark's own label -> hash derivation is unpinned) with their fixed-base tables, and the digest of
 * all of it that every hash of the protocol absorbs.  One-time; outside every timed region (benches/nova.rs:51-58). */
int  vdf_nova_public_params(vdf_ctx* ctx, uint64_t num_iters_per_step, vdf_pp** out);      /* the reference's circuit */
/* The same with the step circuit and the generator family chosen.  gens_family = VDF_GENS_KNOWN_DLOG is for tests only
 * (commitments checkable by the discrete-log identity at full size; such commitments are not binding);
 * VDF_GENS_LABEL_SHAKE derives the generators from the label "vdf-nova-ivc-v1 gens" through SHAKE256, the way nova-snark
 * derives its CommitGens (vdf_bases_generate_label): parameters reproducible from a string. */
int  vdf_nova_public_params_ex(vdf_ctx* ctx, uint64_t num_iters_per_step, int circuit_kind, int gens_family, vdf_pp** out);
/* HBM a parameter set holds, and how to bound the optional part.  Beside the shapes and the generators with their
 * fixed-base tables (2^19 Pallas generators x 16 windows x 64 B = 512 MiB for the reference's circuit at t = 2^16, plus the
 * derived generators of the packed commitment), public_params builds a DIGIT TABLE per side for the generators of the
 * small commitments a step waits on (vdf_bases_precompute_digits: 0.85 MB per generator at a 10-bit window, 2.9 MB at 12):
 * all 10,049 generators of the secondary side and the ~12.5 k of the primary side outside the MinRoot rounds and the early
 * rows.  It is an accelerator only (results are the same group elements without it) and it is BUDGETED: the tables of both
 * sides together take at most vdf_nova_tuning.digit_budget_bytes -- 20 GiB by default, which buys the 10-bit tables
 * (19 GB at t = 2^16); 72 GiB buys the 12-bit ones (65 GB) for ~2.5 % of a step -- and never more than the free HBM less a
 * reserve for what is allocated afterwards (proof buffers, MSM workspaces).  A table that does not fit is skipped
 * (vdf_nova_pp_memory reports it; tuning.verbose logs it).  Flags:
 *   VDF_PP_NO_DIGIT_TABLES  no digit tables: every commitment takes the bucket method (about +0.5 ms per step at t = 2^16)
 *   VDF_PP_NO_EARLY_ROWS    the cross term T of a step is made and committed in one piece instead of ahead of the step
 *                           for the rows that read only the MinRoot rounds (one MSM workspace and one queue fewer) */
enum { VDF_PP_NO_DIGIT_TABLES = 1u, VDF_PP_NO_EARLY_ROWS = 2u };
int  vdf_nova_public_params_flags(vdf_ctx* ctx, uint64_t num_iters_per_step, int circuit_kind, int gens_family, uint32_t flags,
                                  vdf_pp** out);
/* The random oracle as a PARAMETER BLOCK, covered by the parameters' digest.  The reference reaches
 * its RO through nova-snark 0.8.0 -> neptune 7.2.0 (Cargo.toml:14-15), neither of which is in the reference tree: nothing
 * pins their constants here, so this library's default is its own permutation -- and everything the RO is made of is data:
 *   family 0 (VDF_RO_POSEIDON2): Poseidon2-style, width 4, 8 + 56 rounds, constants from SHAKE256 (the default; the only
 *            block of this family the build supports);
 *   family 1 (VDF_RO_POSEIDON):  the original Poseidon permutation -- dense Cauchy MDS matrix 1 / (i + t + j), round
 *            constants from the paper's Grain LFSR, any width 2..25 (a sponge of rate width - 1), even full_rounds <= 16,
 *            partial_rounds <= 128.  vdf_nova_ro_preset(1) is the shape [UPSTREAM-RECALL] neptune gives nova-snark: width 25,
 *            8 + 57 rounds.  Recalled, unpinned, NOT claimed interoperable: it shows that adopting the upstream constants
 *            is a change of this block, not of code.
 * alpha must be 5, challenge_bits 128, hash_bits 250 (what the circuits of this build are written for); any other value is
 * refused.  A block other than the default changes both R1CS shapes and the digest. */
enum { VDF_RO_POSEIDON2 = 0, VDF_RO_POSEIDON = 1 };
typedef struct vdf_nova_ro_params {
  uint32_t struct_size;          /* sizeof(vdf_nova_ro_params) as the caller compiled it */
  int32_t  family, width, full_rounds, partial_rounds, alpha, challenge_bits, hash_bits;
} vdf_nova_ro_params;
int  vdf_nova_ro_preset(int which, vdf_nova_ro_params* out);        /* 0 = this build's default, 1 = the neptune-shaped block */
int  vdf_nova_pp_ro(const vdf_pp* pp, vdf_nova_ro_params* out);     /* the block a parameter set was made under */
/* Everything tunable about a parameter set and the prover that runs over it, in one struct (DESIGN.md says what each
 * choice measured).  vdf_nova_tuning_default fills in the defaults; the environment variables of earlier rounds
 * (VDF_NOVA_*) are read once, by that function's first call, as overrides of those defaults.  A parameter set keeps its
 * copy: two sets in one process may differ. */
typedef struct vdf_nova_tuning {
  uint32_t struct_size;          /* sizeof(vdf_nova_tuning) as the caller compiled it */
  uint32_t flags;                /* VDF_PP_* */
  uint64_t digit_budget_bytes;   /* HBM for the digit tables of both sides together (20 GiB) */
  int32_t  digit_window;         /* 0 = the widest window of 12 .. 8 whose tables fit the budget; 6..12 = exactly this one
                                    (still subject to the free HBM); -1 = none */
  int32_t  early_rows;           /* 2 = the early rows of T start with the step (default), 1 = after the secondary NIFS, 0 = none */
  int32_t  stencil;              /* 1 = the built-in circuits' early rows without the sparse matrices (vdf_nova_pp_stencil) */
  int32_t  small_window;         /* fixed-base window of a side with fewer than 2^17 generators, 6..16 (15) */
  int32_t  big_window;           /* ... with 2^17 and more, 12..20 (16) */
  int32_t  packed_commit;        /* 1 = the reference circuit's rounds committed over derived generators (3t + 4 terms) */
  int32_t  lookahead_early;      /* 1 = the next step's rounds are launched under the primary side's wait */
  int32_t  gate_accumulate;      /* 1 = the lookahead's bucket accumulation is held behind the primary side's direct sum */
  int32_t  fold_on_rows;         /* 1 = the primary fold runs on the early rows' queue */
  int32_t  nifs_ahead;           /* 1 = a step launches the next step's first device phases on its way out */
  int32_t  early_row_parts;      /* 1..3: the early rows as an MSM job of that many parts (1) */
  int32_t  lookahead_priority;   /* wave priority of the lookahead's sort and bucket reduction, 0..3 (1) */
  int32_t  side_accumulate_fill; /* accumulation workgroups per CU the two side queues' MSMs fill, 1..3 (3) */
  int32_t  verbose;              /* 1 = decisions (skipped tables, the window chosen) on stderr */
  int32_t  compress_queues;      /* 1: vdf_nova_compress runs the primary side's two openings on two queues half a round apart (one
                                    opening's sort and bucket reduction under the other's accumulation); 0: in lockstep on one (1) */
  int32_t  rows_at_challenge;    /* 1: the primary fold and the next step's early rows are launched the moment the secondary circuit's
                                    synthesis has derived the fold challenge (a call-back from inside it); 0: after the synthesis (1) */
  int32_t  fold_fused;           /* 1: the stencil kernel of the next step's early rows applies the primary fold to ITS rows of A z, B z,
                                    C z and E on the way (vdf_nifs_cross_term_minroot_fold) and only z and the other rows are folded by a
                                    launch of their own, beside it; 0: one fold over whole vectors in front of the rows (0: measured; needs stencil) */
} vdf_nova_tuning;
void vdf_nova_tuning_default(vdf_nova_tuning* out);
/* public_params with the tuning given (NULL = the defaults); VDF_ERR_BAD_ARG for a field out of range */
int  vdf_nova_public_params_tuned(vdf_ctx* ctx, uint64_t num_iters_per_step, int circuit_kind, int gens_family,
                                  const vdf_nova_tuning* tuning, vdf_pp** out);
int  vdf_nova_pp_tuning(const vdf_pp* pp, vdf_nova_tuning* out);            /* the copy this parameter set runs with */
/* public_params under the given random-oracle block (NULL = the default) and tuning (NULL = the defaults) */
int  vdf_nova_public_params_ro(vdf_ctx* ctx, uint64_t num_iters_per_step, int circuit_kind, int gens_family,
                               const vdf_nova_ro_params* ro, const vdf_nova_tuning* tuning, vdf_pp** out);
/* wall-clock milliseconds of the stages of the call that made `pp`: [0] both shapes + digest (host), [1] shapes to the
 * device, [2] generators, [3] fixed-base tables (the packed commitment's included), [4] digit tables, [5] the rest, [6] total */
int  vdf_nova_pp_setup_ms(const vdf_pp* pp, double ms[7]);
/* bytes of HBM held per side: generators, their fixed-base table, the digit table (0 = none; *skipped bit s set when
 * side s wanted one and it did not fit).  The primary side's figures include the derived generators of the packed
 * commitment and their table.  (Not in these figures: after the first vdf_nova_compress / vdf_nova_verify_compressed a set also
 * keeps that call's scratch vectors, 0.18 GB for the reference's circuit at t = 2^16, until it is freed.) */
int  vdf_nova_pp_memory(const vdf_pp* pp, uint64_t gens_bytes[2], uint64_t table_bytes[2], uint64_t digit_bytes[2], uint32_t* skipped);
void vdf_nova_pp_free(vdf_pp* pp);
int  vdf_nova_pp_sizes(const vdf_pp* pp, int side, uint64_t* num_cons, uint64_t* num_vars, uint64_t* num_io, uint64_t* nnz3,
                       uint64_t* num_gens);
/* the 250-bit digest of the parameters (little-endian), and the run of primary variables the GPU fills (MinRoot rounds) */
int  vdf_nova_pp_digest(const vdf_pp* pp, uint8_t out[32]);
int  vdf_nova_pp_segment(const vdf_pp* pp, uint64_t* begin, uint64_t* len);
/* the run of primary constraints whose share of a step's cross term T and of comm_T prove_step makes ahead of the rest of
 * the step (they read only that segment, the step's input and the constant); len = 0: none, T is committed in one piece */
int  vdf_nova_pp_early_rows(const vdf_pp* pp, uint64_t* begin, uint64_t* len);
/* 4 / 3: the early rows are the built-in MinRoot stencil (the reference's rounds / the bound form), verified against the
 * shape's triples when the parameters were made -- their cross term runs without the sparse matrices
 * (vdf_hip.h vdf_nifs_cross_term_minroot); 0: they run through the generic sparse kernel (a custom circuit, or no early rows) */
int  vdf_nova_pp_stencil(const vdf_pp* pp);
/* The same answer without a device (host only): builds the shape of the built-in step circuit at t, finds the early rows
 * and compares them with the stencil; returns 4 / 3 / 0 (negative: an error code).  Outputs may be NULL. */
int  vdf_nova_shape_stencil(uint64_t t, int circuit_kind, uint64_t* early_begin, uint64_t* early_len, uint64_t* seg_begin);

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
/* One RecursiveSNARK::prove_step (:342-349): *proof == NULL starts a new proof (the `None` case).  z0_secondary is
 * [0] (:310, :389-391).  What a step computes that does not depend on the chain -- the MinRoot rounds of circuit k and
 * their share of the commitment -- is enqueued one call early, for circuit k + 1 of the same `circuits`, on a second
 * context the proof owns; a call for any other step finds no such work waiting and does it then.  The results are
 * those of a prover without lookahead, and no call returns while anything in flight still reads the circuits' memory. */
int  vdf_nova_prove_step(vdf_pp* pp, vdf_proof** proof, const vdf_circuits* circuits, size_t k, const vdf_fe z0[3]);
/* NovaVDFProof::verify(pp, num_steps, z0, zi), :370-387: *ok = 1 iff the proof is valid for num_steps steps from z0
 * (two output hashes, three satisfiability claims), the verified zi_primary equals zi, and zi_secondary == [0]
 * (the Ok(bool) of :386). */
int  vdf_nova_verify(const vdf_proof* proof, vdf_pp* pp, size_t num_steps, const vdf_fe z0[3], const vdf_fe zi[3], int* ok);
void vdf_nova_proof_free(vdf_proof* proof);
size_t vdf_nova_proof_num_steps(const vdf_proof* proof);
/* Introspection for the parity tests.  Instances: commitments as affine points, u and X[2] in Montgomery form of the
 * instance's own scalar field (Fq on the primary side, Fp on the secondary).  Witness pointers: device memory,
 * z = [W | u | X] (W = the first num_vars elements) and E (NULL for the fresh instance). */
enum { VDF_INST_RUNNING_PRIMARY = 0, VDF_INST_RUNNING_SECONDARY = 1, VDF_INST_FRESH_SECONDARY = 2,
       /* witness pointer only: z of the fresh primary instance the LAST prove_step folded (its instance: vdf_nova_proof_last_step);
        * valid until the next step */
       VDF_INST_FRESH_PRIMARY_LAST = 3 };
int  vdf_nova_proof_instance(const vdf_proof* proof, int which, vdf_affine* comm_W, vdf_affine* comm_E, vdf_fe* u, vdf_fe X[2]);
int  vdf_nova_proof_witness_ptrs(const vdf_proof* proof, int which, const void** d_z, const void** d_E);
int  vdf_nova_proof_zi(const vdf_proof* proof, vdf_fe zi_primary[3], vdf_fe zi_secondary[1]);
/* by-products of the last prove_step: the fresh primary instance, both cross-term commitments (identity in the base
 * step) and both fold challenges (canonical 128-bit integers, little-endian limbs) */
typedef struct { vdf_affine comm_W1; vdf_fe X1[2]; vdf_affine comm_T1, comm_T2; uint64_t r1[4], r2[4]; } vdf_nova_step_info;
int  vdf_nova_proof_last_step(const vdf_proof* proof, vdf_nova_step_info* out);
/* host wall-clock of the last prove_step, milliseconds: [0] enqueue + wait of the secondary cross term and commitments,
 * [1] synthesis of the primary augmented circuit, [2] upload + enqueue of the primary cross term and commitments,
 * [3] wait for them, [4] synthesis of the secondary augmented circuit, [5] upload + fold launches, [6] lookahead
 * launch, [7] total. */
int  vdf_nova_last_step_ms(const vdf_proof* proof, double ms[8]);
/* Per-launch timing of everything a prover enqueues (vdf_hip.h vdf_ctx_kernel_events) over its three queues -- queue[i] =
 * 0 the step's chain (the parameters' context), 1 the lookahead (MinRoot rounds of the next step and their commitment),
 * 2 the early rows of the cross term -- on one time line.  For bench.py's prove_step.roofline; measure rates with it off. */
int  vdf_nova_proof_set_kernel_timing(vdf_proof* proof, int enable);
int  vdf_nova_proof_kernel_events(vdf_proof* proof, vdf_kernel_event* out, int* queue, size_t cap, size_t* n);

/* ---- the step-circuit seam (src/nova/proof.rs:79-153: `impl StepCircuit for InverseMinRootCircuit` -- arity, synthesize,
 * output) ---------------------------------------------------------------------------------------------------------
 * A step circuit written by the host: `synthesize` receives the constraint system and the arity handles of z_in, makes
 * its variables and constraints through the vdf_cs_* calls below, and stores the handles of z_out.  The same function
 * runs in two modes, as a bellperson circuit does: when the parameters are made the constraints are recorded (values
 * are ignored); in every prove_step only the values are computed (vdf_cs_is_witness).  `output` of the trait is the
 * value of z_out after a witness synthesis.  The library's own circuits (both MinRoot forms, TrivialTestCircuit) sit
 * behind the same C++ interface (host/r1cs.hpp StepCircuit); a custom circuit's variables are all made on the host. */
typedef struct vdf_cs vdf_cs;
typedef uint32_t vdf_num;                 /* a linear combination with its value, owned by the vdf_cs */
typedef struct {
  size_t arity;
  int (*synthesize)(void* self, vdf_cs* cs, const vdf_num* z_in, vdf_num* z_out);       /* 0 = ok */
  void* self;
} vdf_step_circuit;
int     vdf_cs_is_witness(const vdf_cs* cs);                          /* 1: values are live; 0: the shape is being recorded */
vdf_num vdf_cs_const(vdf_cs* cs, const vdf_fe* k);                    /* the constant k (Montgomery form) */
vdf_num vdf_cs_add(vdf_cs* cs, vdf_num a, vdf_num b);
vdf_num vdf_cs_sub(vdf_cs* cs, vdf_num a, vdf_num b);
vdf_num vdf_cs_scale(vdf_cs* cs, vdf_num a, const vdf_fe* k);
vdf_num vdf_cs_alloc(vdf_cs* cs, const vdf_fe* value);                /* a new variable (AllocatedNum::alloc) */
vdf_num vdf_cs_mul(vdf_cs* cs, vdf_num a, vdf_num b);                 /* a new variable a * b and its constraint */
int     vdf_cs_enforce(vdf_cs* cs, vdf_num a, vdf_num b, vdf_num c);  /* a * b = c */
int     vdf_cs_value(const vdf_cs* cs, vdf_num a, vdf_fe* out);       /* witness mode */
/* public_params / prove_step / verify for a custom primary step circuit (the secondary stays TrivialTestCircuit).
 * z0, zi: `arity` elements.  compress, verify_compressed and the wire formats work on such proofs unchanged. */
int  vdf_nova_public_params_custom(vdf_ctx* ctx, const vdf_step_circuit* primary, int gens_family, vdf_pp** out);
int  vdf_nova_prove_step_custom(vdf_pp* pp, vdf_proof** proof, const vdf_step_circuit* primary, const vdf_fe* z0);
int  vdf_nova_verify_custom(const vdf_proof* proof, vdf_pp* pp, size_t num_steps, const vdf_fe* z0, const vdf_fe* zi, int* ok);

/* ---- host-only entry points (no device): what the CPU tests pin against oracle/nova.py ------------------------- */
/* the random oracle: lane 1 of the sponge after absorbing xs under `tag` (a full field element, Montgomery in and out) */
int  vdf_nova_ro_hash(int field, uint64_t tag, const vdf_fe* xs, size_t n, vdf_fe* out);
/* digest of the parameters public_params would make (both shapes synthesised on the host), and the sizes per side:
 * sizes[side] = {num_cons, num_vars, nnz(A) + nnz(B) + nnz(C)} */
int  vdf_nova_shape_digest(uint64_t num_iters_per_step, int circuit_kind, int gens_family, uint8_t out[32], uint64_t sizes[2][3]);
/* the same shape as COO triples per matrix (A, B, C), in the order the constraints were made, values in Montgomery form of the
 * side's scalar field: call once with rows = cols = vals = NULL for nnz[3], then with arrays of those lengths. */
int  vdf_nova_shape_export(uint64_t num_iters_per_step, int circuit_kind, int side, uint64_t nnz[3], uint32_t* const rows[3],
                           uint32_t* const cols[3], vdf_fe* const vals[3]);
int  vdf_nova_shape_digest_custom(const vdf_step_circuit* primary, int gens_family, uint8_t out[32], uint64_t sizes[2][3]);
/* One augmented circuit synthesised on the host with every variable computed there (small t only).  The inputs that
 * belong to the folded side (U_u, U_X, u_X) are in Montgomery form of THAT side's scalar field, everything else in the
 * circuit's own field.  result / input: the MinRoot step's states (side 0; ignored for side 1).  arity = 3 / 1. */
typedef struct {
  vdf_fe params, i, z0[3], zi[3];
  vdf_affine U_comm_W, U_comm_E; vdf_fe U_u, U_X[2];
  vdf_affine u_comm_W; vdf_fe u_X[2];
  vdf_affine T;
} vdf_nova_aug_inputs;
int  vdf_nova_aug_synthesize(int side, uint64_t num_iters_per_step, int circuit_kind, const vdf_nova_aug_inputs* in,
                             const vdf_state* result, const vdf_state* input, vdf_fe* W, size_t w_cap, size_t* num_vars,
                             size_t* num_cons, vdf_fe X[2], vdf_fe z_next[3]);
/* the host-only entry points under another random-oracle block (NULL = the default): sponge, shape digest, synthesis */
int  vdf_nova_ro_hash_ro(const vdf_nova_ro_params* ro, int field, uint64_t tag, const vdf_fe* xs, size_t n, vdf_fe* out);
int  vdf_nova_shape_digest_ro(const vdf_nova_ro_params* ro, uint64_t num_iters_per_step, int circuit_kind, int gens_family, uint8_t out[32],
                              uint64_t sizes[2][3]);
int  vdf_nova_aug_synthesize_ro(const vdf_nova_ro_params* ro, int side, uint64_t num_iters_per_step, int circuit_kind,
                                const vdf_nova_aug_inputs* in, const vdf_state* result, const vdf_state* input, vdf_fe* W, size_t w_cap,
                                size_t* num_vars, size_t* num_cons, vdf_fe X[2], vdf_fe z_next[3]);

/* of the calling thread's last augmented-circuit synthesis: how many slope inverses the native pre-pass queued (batched
 * inversion) and how many of them were wrong or unused (0 for well-formed inputs: the queue is only an accelerator) */
int  vdf_nova_synthesis_stats(uint64_t* queued, uint64_t* misses);

/* ---- compression (src/nova/proof.rs:360-368, :383) ---------------------------------------------------------
 * NovaVDFProof::compress -> nova-snark CompressedSNARK::prove: the last secondary instance is folded into the running
 * one, then one succinct argument PER SIDE (SS1 / SS2 of :32-33) that the running primary instance and the folded
 * secondary instance are satisfiable, instead of their witnesses.  Protocol "vdf-spartan-v3" (oracle/spartan.py): a
 * Spartan-style sum-check argument with inner-product-argument openings under the side's own Pedersen generators; the
 * extra generator of the openings is generator number num_gens of the same family.  Self-consistent, bit-exact against
 * the oracle, not interchangeable with nova-snark (unpinned).  Every pass over a vector runs on the GPU. */
int  vdf_nova_compress(const vdf_proof* proof, vdf_pp* pp, vdf_snark** out);
/* NovaVDFProof::verify for the Compressed variant (:383): *ok = 1 iff the two output hashes match the carried
 * instances for num_steps steps from z0, both arguments verify (the secondary one for the instance the verifier
 * folds itself), the carried zi_primary equals zi and zi_secondary == [0]. */
int  vdf_nova_verify_compressed(const vdf_snark* snark, vdf_pp* pp, size_t num_steps, const vdf_fe z0[3], const vdf_fe zi[3],
                                int* ok);
void vdf_nova_snark_free(vdf_snark* snark);
/* Flat canonical encoding of the two arguments, primary then secondary (little-endian, non-Montgomery, 64-byte points;
 * layout in the implementation and in oracle/wire.py): size, export, and import -- which replaces the arguments of
 * `snark` and returns VDF_ERR_NONCANONICAL for an out-of-range field element or a point off its curve. */
size_t vdf_nova_snark_size(const vdf_snark* snark);
int  vdf_nova_snark_bytes(const vdf_snark* snark, uint8_t* out, size_t cap);
int  vdf_nova_snark_set_bytes(vdf_snark* snark, const uint8_t* in, size_t len);

/* ---- wire formats ---------------------------------------------------------------------
 * The reference keeps its proofs in memory only (src/nova/proof.rs:52-55 derives no serialisation); these encodings
 * are this library's own, versioned by their magic.  Field elements: 32 bytes, canonical, little-endian.  Points:
 * 32 bytes, canonical little-endian x with the parity of y in bit 255, the identity as 32 zero bytes.
 *
 *   relaxed instance = comm_W [32] | comm_E [32] | u [32] | X [2 x 32];   strict instance = comm_W [32] | X [2 x 32]
 *   "VDFSNK03" compressed proof = magic[8] | t u64 | digest of the public parameters [32]
 *        | running primary instance | running secondary instance | last secondary instance (strict)
 *        | cross-term commitment of the last fold [32] | z_i primary [96] | z_i secondary [32]
 *        | argument of the primary side | argument of the secondary side   (as vdf_nova_snark_bytes, 32-byte points)
 *   "VDFRSK02" running proof    = magic[8] | t u64 | steps u64 | digest [32] | z_0 [96] | z_i primary [96] | z_i secondary [32]
 *        | the same three instances | W1 | E1 | W2 | E2 | w2
 *
 * Both are constant-size in the number of steps.  Deserialisation fails with VDF_ERR_BAD_ARG for a foreign magic or
 * other public parameters, VDF_ERR_BAD_LENGTH for a length that does not fit the shapes, and VDF_ERR_NONCANONICAL for
 * an out-of-range field element or bytes that decode to no curve point.  The running proof is a checkpoint:
 * vdf_nova_proof_deserialize rebuilds the device-resident state (including A z, B z, C z of the running instances),
 * refuses witnesses that do not open their commitments, and vdf_nova_prove_step continues from it. */
/* The 32-byte point encoding by itself (host arithmetic only, no device); curve = VDF_CURVE_PALLAS / VDF_CURVE_VESTA,
 * coordinates in Montgomery form like everywhere in this ABI.  decompress: VDF_ERR_NONCANONICAL unless the bytes are
 * exactly what compress writes for some point. */
int  vdf_nova_point_compress(int curve, const vdf_affine* p, uint8_t out[32]);
int  vdf_nova_point_decompress(int curve, const uint8_t in[32], vdf_affine* out);
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
