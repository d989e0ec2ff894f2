// Constraint system, gadgets and circuits of the Nova layer (libvdf_nova.so): the host-side counterpart of what the
// reference reaches through bellperson 0.22 (ConstraintSystem / AllocatedNum / LinearCombination, src/nova/proof.rs:3-9)
// and nova-snark 0.8.0's augmented circuit (Cargo.toml:15).  Neither crate's source is in /root/reference; the
// specification this file implements line by line is oracle/nova.py (protocol "vdf-nova-ivc-v1", parity unpinned
// against nova-snark, pinned bit-for-bit against that restatement by tests/test_nova_host.py and tests/test_gpu_nova.py).
//
// One code path serves both uses of a circuit: SHAPE mode records the R1CS matrices (public_params, once), WITNESS
// mode only computes the variable assignment (every prove_step) -- linear combinations are then empty and cost nothing.
#pragma once
#include <cstdlib>
#include <functional>
#include <memory>
#include <vector>
#include "host_math.hpp"

namespace vdfnova {
using namespace vdfhost;

// the one place libvdf_nova.so reads the environment: tuning overrides (default_tuning, once) and two debugging switches of
// the synthesis (VDF_NOVA_SYNTH_TRACE, VDF_NOVA_SEQ_SYNTH), each read once into a static
inline const char* env_override(const char* name) { return std::getenv(name); }


// variable keys: W index k -> k; the constant column -> KEY_ONE; public IO k -> KEY_ONE + 1 + k.  Ascending key order is
// the column order of z = (W, u, X).
constexpr uint32_t KEY_ONE = 0x40000000u;
struct Term { uint32_t key; Fe c; };
using LC = std::vector<Term>;
struct Num { Fe v; LC lc; };

struct Coo { std::vector<uint32_t> rows, cols; std::vector<Fe> vals; };

// Witness vectors are ~0.5 MB: taken from a small pool and given back (a fresh allocation of that size is an mmap and a
// page fault per 4 KB touched, every step -- a third of a synthesis).
std::vector<Fe> witness_buffer_take();
void witness_buffer_give(std::vector<Fe>&& v);

// ---- the random oracle as a parameter block (include/vdf_nova.h vdf_nova_ro_params; oracle/poseidon.py RoSpec) -----
// family 0: this build's Poseidon2-style permutation (width 4, 8 + 56 rounds, SHAKE256 constants: the fast path below);
// family 1: the original Poseidon permutation [UPSTREAM-RECALL: neptune's shape] -- dense Cauchy MDS, Grain-LFSR constants,
// any width up to 25, a sponge of rate width - 1.  The block is covered by the parameters' digest (digest_shapes absorbs
// its label, and the shapes change with it); the two truncations are recorded in it and checked against what this build
// supports (128 / 250).
constexpr int RO_MAX_T = 25;
struct RoSpec {
  int family = 0, width = 4, full_rounds = 8, partial_rounds = 56, alpha = 5, challenge_bits = 128, hash_bits = 250;
  bool operator==(const RoSpec& o) const {
    return family == o.family && width == o.width && full_rounds == o.full_rounds && partial_rounds == o.partial_rounds &&
           alpha == o.alpha && challenge_bits == o.challenge_bits && hash_bits == o.hash_bits;
  }
};
struct RoInstance {
  RoSpec spec;
  int rate = 3;
  std::vector<Fe> rc[2], mds[2];         // family 1, per field id: (RF + RP) x width round constants, width x width matrix
  std::vector<uint8_t> label;            // what the parameters' digest absorbs for the RO
  bool is_default = true;
};
const RoInstance* ro_default();
const RoInstance* ro_instance(const RoSpec& s);      // null: a block this build does not support; instances live for the process

class CS {
 public:
  CS(int field_id, bool shape_mode, const RoInstance* ro = nullptr);      // ro = null: the default block
  const RoInstance* const ro;
  ~CS();
  CS(const CS&) = delete;
  CS& operator=(const CS&) = delete;
  const int field_id;
  const Field& F;
  const bool shape;                      // true: record constraints; false: witness only
  std::vector<Fe> W, X;
  size_t rows = 0;
  // Witness mode: a run of dev_len variables that some other party fills (the MinRoot rounds, on the GPU) starts at
  // variable index dev_begin; W then holds only the host-made values, packed: W[0, dev_begin) are variables
  // [0, dev_begin), W[dev_begin, ...) are variables [dev_begin + dev_len, ...).
  size_t dev_begin = 0, dev_len = 0;
  // variables [step_begin, step_end) are the step circuit's own (set by synthesize_augmented)
  size_t step_begin = 0, step_end = 0;
  size_t num_vars() const { return W.size() + dev_len; }

  Num constant(const Fe& k) const;
  Num constant_u64(uint64_t k) const { return constant(from_u64(k, F)); }
  Num zero_num() const { Num n; n.v = vdfhost::zero(); return n; }
  Num add(const Num& a, const Num& b) const;
  Num sub(const Num& a, const Num& b) const;
  Num scale(const Num& a, const Fe& k) const;
  Num scale_small(const Num& a, unsigned k) const;          // k <= 16: the value by additions
  Num alloc(const Fe& v);
  Num alloc_io(const Fe& v);
  void skip(size_t n, size_t cons);                          // n variables (with their `cons` constraints) left to the device (witness mode, once)
  void enforce(const Num& a, const Num& b, const Num& c);
  Num mul(const Num& a, const Num& b);
  void enforce_equal(const Num& a, const Num& b);
  // COO triples in the column numbering of z = (W, u, X); call once, after synthesis (shape mode)
  void finish(Coo out[3]) const;

  // ---- inversions.  A witness holds ~530 field inverses (curve slopes, is-zero helpers); at 6 us each they would be
  // the whole cost of a step.  Two mechanisms keep them to a handful of real inversions per circuit:
  //  * slopes: a native pre-pass (projective arithmetic, then batched inversion) queues the inverses in the order the
  //    gadgets will ask for them; take_inverse hands the next one out after checking den * inv = 1, and computes the
  //    inverse itself if the queue is empty or the check fails -- the queue is an accelerator, never a source of truth;
  //  * is-zero helpers are used by no later computation: alloc_inverse_later reserves the variable and resolve() fills
  //    all of them with one batched inversion at the end of the synthesis.
  std::vector<Fe> inv_queue;
  size_t inv_pos = 0, inv_misses = 0;
  Fe take_inverse(const Fe& den);
  Num alloc_inverse_later(const Fe& a);
  void resolve();

 private:
  struct Row { LC a, b, c; };
  std::vector<Row> cons_;
  std::vector<std::pair<size_t, Fe>> later_;       // (position in W, value to invert)
};

// ---- the random oracle: Poseidon2-style permutation, width 4 (oracle/poseidon.py) -----------------------------------
constexpr int RO_T = 4, RO_RATE = 3, RO_RF = 8, RO_RP = 56;
struct RoConstants { Fe ext[RO_RF][RO_T]; Fe in[RO_RP]; unsigned mu_minus_1[RO_T]; };
const RoConstants& ro_constants(int field_id);
void ro_permute(Fe* s, int field_id, const RoInstance* ro = nullptr);      // s: ro->spec.width elements (4 for the default)
Fe ro_hash(int field_id, uint64_t tag, const Fe* xs, size_t n, const RoInstance* ro = nullptr);       // full field element (lane 1)

// ---- gadgets (allocation and constraint order as in oracle/nova.py) -------------------------------------------------
Num is_zero(CS& cs, const Num& a);
Num select(CS& cs, const Num& cond, const Num& a, const Num& b);
std::vector<Num> alloc_bits(CS& cs, const uint64_t v[4], int n);      // v = the canonical integer, little-endian limbs
Num pack(const CS& cs, const Num* bits, size_t n);
std::vector<Num> strict_bits(CS& cs, const Num& a);
Num poseidon_hash(CS& cs, uint64_t tag, const std::vector<Num>& xs);
void check_on_curve(CS& cs, const Num& x, const Num& y, const Num& inf);
void ec_scalar_mul(CS& cs, const std::vector<Num>& bits, const Num& px, const Num& py, const Num& p_inf, Num* rx, Num* ry);
void ec_add_complete(CS& cs, const Num& x1, const Num& y1, const Num& x2, const Num& y2, Num* ox, Num* oy);
// n inverses with one inversion (Montgomery's trick); zeros stay zero
void batch_inverse(Fe* v, size_t n, const Field& F);
// the slope inverses of  U + [r] P  as ec_scalar_mul followed by ec_add_complete will request them, appended to out
// (prepared: the doubling side, which needs no r and may be made ahead of it -- the points 2^k P, the tangents' inverses,
// the affine doublings and the tangent's witness values; null: made here)
struct FoldPre {
  std::vector<Pt> w;
  std::vector<Fe> tan_inv, wx, wy, x2, lam;
  size_t misses = 0;
};
void ec_fold_prepare(const Field& F, const Aff& P, int bits, FoldPre* pre);
void ec_fold_inverses(const Field& F, const Aff& U, const Aff& P, const uint64_t r[4], int bits, std::vector<Fe>* out,
                      const FoldPre* prepared = nullptr);
void fold_foreign(CS& cs, const Num& a_lo, const Num& a_hi, const std::vector<Num>& b_bits, const std::vector<Num>& r_bits,
                  const Field& foreign, Num* r_lo, Num* r_hi);

// ---- step circuits: the seam of src/nova/proof.rs:79-153 (trait StepCircuit: arity / synthesize / output) ----------
struct StepCircuit {
  virtual ~StepCircuit() {}
  virtual size_t arity() const = 0;
  // allocates the circuit's variables and constraints over z_in, returns z_out (values valid in witness mode)
  virtual std::vector<Num> synthesize(CS& cs, const std::vector<Num>& z) const = 0;
  virtual void output(const Fe* z, Fe* out) const = 0;
  // true: output() tells z_out without a synthesis (the augmented circuit then starts hashing it before the folds are done)
  virtual bool output_known() const { return false; }
};

struct MinRootState { Fe x, y, i; };
// InverseMinRootCircuit (src/nova/proof.rs:57-230).  bound = false: the reference's circuit exactly (4 variables per
// round, new_x allocated at :167-173 and used by no constraint); bound = true (what the product proves by default):
// new_x is the linear combination y - i + 1 itself (3 variables per round, the same three constraints per round).
struct InverseMinRootCircuit : StepCircuit {
  uint64_t t = 0;
  bool bound = true, blank = true;
  bool device_rounds = false;            // witness mode: leave the per-round variables to the GPU kernel (cs.skip)
  MinRootState result, input;
  size_t arity() const override { return 3; }
  std::vector<Num> synthesize(CS& cs, const std::vector<Num>& z) const override;
  void output(const Fe* z, Fe* out) const override;
  bool output_known() const override { return !blank; }
  size_t vars_per_round() const { return bound ? 3 : 4; }
};
// nova-snark's TrivialTestCircuit (src/nova/proof.rs:258-260): arity 1, z_out = z_in
struct TrivialTestCircuit : StepCircuit {
  size_t arity() const override { return 1; }
  std::vector<Num> synthesize(CS&, const std::vector<Num>& z) const override { return z; }
  void output(const Fe* z, Fe* out) const override { out[0] = z[0]; }
  bool output_known() const override { return true; }
};

// ---- instances and the augmented circuit ---------------------------------------------------------------------------
constexpr int HASH_BITS = 250, CHAL_BITS = 128, LIMB_BITS = 126, AUG_IO = 2;
constexpr uint64_t TAG_STATE = 1, TAG_CHAL = 2;

// a running (relaxed) instance of one side, as the OTHER side's circuit sees it: everything native to that circuit's
// field except u and X, which are integers below the instance's own scalar modulus
struct RelaxedInst {
  Aff comm_W, comm_E;                    // coordinates: Montgomery form in the base field of the instance's curve
  uint64_t u[4], X[2][4];                // canonical integers (u stays far below both moduli)
};
struct AugInputs {
  Fe params, i;                          // Montgomery form in the circuit's field
  std::vector<Fe> z0, zi;
  RelaxedInst U;
  Aff u_W;
  uint64_t u_X[2][4];                    // canonical integers (250-bit hashes)
  Aff T;
  const RoInstance* ro = nullptr;        // the random oracle's parameter block (null: the default)
  // witness mode, the block-wise synthesis only: called on the calling thread the moment the fold challenge r is derived
  // (a few tens of microseconds into the late half, long before the witness is assembled) -- a prover launches what needs
  // nothing but r from here.  Not called by the sequential synthesis or in shape mode: a caller checks whether it ran.
  std::function<void(const uint64_t r[4])> on_challenge;
};
void relaxed_elements(const RelaxedInst& U, const Field& F, Fe out[9]);       // what a running instance is hashed as
Fe hash_state(int field_id, const Fe& params, const Fe& i, const std::vector<Fe>& z0, const std::vector<Fe>& zi,
              const RelaxedInst& U, uint64_t out_int[4], const RoInstance* ro = nullptr);
void hash_challenge(int field_id, const Fe& params, const RelaxedInst& U, const Aff& u_W, const uint64_t u_X[2][4], const Aff& T,
                    uint64_t r_out[4], const RoInstance* ro = nullptr);
// side 0 = primary (circuit over Fq, folds Vesta instances), side 1 = secondary; returns z_{i+1}
// unew (optional): the nine elements of the running instance the circuit hands on (the folded one, or the base case's);
// r (optional): the fold challenge it derived, a 128-bit integer
// Witness mode, in two halves.  Everything that does not depend on the two commitments a step is waiting for (u.comm_W
// and T) -- the state hash, the challenge hash up to the running instance, the bits of u.X, the output hash up to z_out
// -- can be made while the device still computes them: synthesize_augmented_early takes the inputs with u_W and T unset
// and returns what it prepared; synthesize_augmented(..., early) then only finishes.  Without `early` the same work
// happens inside the call.  The result is the same witness either way.
struct AugEarly;
void aug_early_free(AugEarly* e);
typedef std::unique_ptr<AugEarly, void (*)(AugEarly*)> AugEarlyPtr;
AugEarlyPtr synthesize_augmented_early(int side, const AugInputs& in, const StepCircuit& step);
std::vector<Fe> synthesize_augmented(CS& cs, int side, const AugInputs& in, const StepCircuit& step, Fe* unew = nullptr,
                                     uint64_t* r = nullptr, AugEarly* early = nullptr);
// of the calling thread's last synthesize_augmented in witness mode: slope inverses queued by the pre-pass, and how many of
// them were wrong or left over (0 unless the inputs were malformed)
void last_synthesis_stats(uint64_t* queued, uint64_t* misses);
inline int side_field(int side) { return side == 0 ? VDF_FIELD_FQ : VDF_FIELD_FP; }
inline int side_curve(int side) { return side == 0 ? VDF_CURVE_PALLAS : VDF_CURVE_VESTA; }

// 256-bit helpers on canonical little-endian limbs
inline void fe_to_int(const Fe& mont, const Field& F, uint64_t out[4]) { const Fe c = from_mont(mont, F); memcpy(out, c.l, 32); }
inline Fe int_to_fe(const uint64_t v[4], const Field& F) {      // v < 2^256, reduced
  Fe c;
  memcpy(c.l, v, 32);
  while (geq(c.l, F.m)) sub4(c.l, F.m);
  return to_mont(c, F);
}

}  // namespace vdfnova
