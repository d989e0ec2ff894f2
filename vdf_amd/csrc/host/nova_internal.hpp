// Shared declarations of the translation units of libvdf_nova.so (minroot_host.cpp, r1cs.cpp, nova_host.cpp,
// compress_host.cpp, wire_host.cpp): error plumbing, the two sides of the curve cycle, the handle structs.
#pragma once
#include <array>
#include <chrono>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <mutex>
#include <vector>
#include "../../../include/vdf_nova.h"
#include "host_math.hpp"
#include "r1cs.hpp"

namespace vdfnova {
using namespace vdfhost;

int fail(int code, const std::string& msg);          // records the message for vdf_nova_last_error, returns code
// no C++ exception crosses the C ABI: an entry point that allocates or synthesises runs its body through this
template <class F>
inline int nova_guard(F&& body) {
  try { return body(); }
  catch (const std::bad_alloc&) { return fail(VDF_ERR_OOM, "host allocation failed"); }
  catch (const std::exception& ex) { return fail(VDF_ERR_DEVICE, ex.what()); }
  catch (...) { return fail(VDF_ERR_DEVICE, "unknown failure"); }
}
#define HIPCALL(ctx, expr)                                                          \
  do {                                                                              \
    int rc__ = (expr);                                                              \
    if (rc__ != VDF_OK) return ::vdfnova::fail(rc__, std::string(#expr) + ": " + vdf_last_error(ctx)); \
  } while (0)

inline double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- MinRoot (minroot_host.cpp) -------------------------------------------------------------------------
struct St { Fe x, y, i; };
inline St load_state(const vdf_state* s) { St r; memcpy(&r, s, sizeof(St)); return r; }
inline void store_state(vdf_state* o, const St& s) { memcpy(o, &s, sizeof(St)); }
inline bool valid_field(int f) { return f == VDF_FIELD_FP || f == VDF_FIELD_FQ; }
inline bool valid_mode(int m) { return m >= 0 && m <= 3; }

// ---- Nova (nova_host.cpp) ---------------------------------------------------------------------------------
constexpr int NUM_IO = AUG_IO;             // public IO of an augmented circuit
constexpr uint64_t GENS_SEED = 0x4e6f7661; // "Nova": label of the generator families
constexpr int PRIMARY = 0, SECONDARY = 1;  // G1 = Pallas / G2 = Vesta, src/nova/proof.rs:26-27

// One side of the cycle: an augmented circuit over F (the scalar field of `curve`), its R1CS shape and the Pedersen
// generators its witnesses are committed under -- everything the folding, the satisfiability check and the
// compression argument need to know about "the instance's side".
// Scratch vectors of the compression SNARK, one block per side, kept by the parameter set from one call to the next (twelve
// device allocations and as many frees per vdf_nova_compress were 0.6 ms of its 27)
struct Arena { void* p = nullptr; size_t cap = 0; };
struct Side {
  int side = 0, field = 0, curve = 0;
  const Field* F = nullptr;                // scalars of this side's instances (= the circuit's field)
  const Field* Fb = nullptr;               // coordinates of this side's commitments
  vdf_ctx* ctx = nullptr;
  Arena* arena = nullptr;                  // the parameter set's scratch block for this side (compress / verify_compressed)
  vdf_ctx* ctx_b = nullptr;                // compress only (a copy of the side made for one call): a second queue for the E opening
  size_t num_cons = 0, num_vars = 0, ncols = 0, nnz3 = 0, num_gens = 0;
  vdf_shape* shape = nullptr;
  vdf_bases* gens = nullptr;
  Aff gen_u;                               // the extra generator of the inner-product arguments: number num_gens of the family
  void* d_zero = nullptr;                  // num_cons zero elements (satisfiability residual)
  uint8_t digest[32];                      // the parameters' digest (same on both sides; bound into the argument's transcript)
};

// instance of one side; u and X in Montgomery form of that side's own field
struct Inst { Aff comm_W, comm_E; Fe u; Fe X[NUM_IO]; };
RelaxedInst to_relaxed(const Inst& in, const Field& own);
// relaxed satisfiability of (inst, z = [W | u | X], E) on `sd`: commitments open and A z o B z = u C z + E
int check_sat(const Side& sd, const Inst& in, const void* d_z, const void* d_E, void* d_scratch_abc[3], void* d_scratch_T, bool* ok);

}  // namespace vdfnova

using vdfhost::Aff; using vdfhost::Fe;
using vdfnova::NUM_IO;

struct vdf_pp {
  vdf_ctx* ctx = nullptr;
  uint64_t t = 0;
  int circuit_kind = 0, gens_family = 0;
  vdfnova::Side s[2];
  uint8_t digest[32];                      // 250-bit little-endian integer
  Fe params[2];                            // the digest as an element of each side's field
  // variables of the primary witness that the GPU fills from the forward trace (the MinRoot rounds): [seg_begin, seg_begin + seg_len)
  size_t seg_begin = 0, seg_len = 0;
  // constraints of the primary shape that read nothing of a fresh witness but that segment (and the constant): their
  // share of a step's cross term and of its commitment is made ahead of the rest, [ahead_row, ahead_row + ahead_rows)
  size_t ahead_row = 0, ahead_rows = 0;
  int stencil_per = 0;           // 3 / 4: the early rows are the built-in MinRoot stencil with that many variables per round, checked against
                                 // the shape at public_params -- their cross term needs no sparse matrix (vdf_nifs_cross_term_minroot); 0: generic rows
  int ahead_mode = 2;            // when they run: 2 = from the start of the step, beside the secondary side's NIFS; 1 = after it (tuning)
  size_t arity = 3;                        // of the primary step circuit (z0, zi)
  // the reference's step circuit only: generators of the packed commitment to the MinRoot rounds (3t + 4 points derived from
  // the 4t + 1 of the segment: vdf_hip.h vdf_minroot_step_segment_packed), with a fixed-base table of their own
  vdf_bases* seg_gens = nullptr;
  const vdfnova::RoInstance* ro = nullptr;  // the random oracle's parameter block (covered by the digest); never null once the set is made
  vdf_ctx* aux_ctx = nullptr;              // a second queue of the same device for compress (the secondary side's argument runs beside the
                                           // primary's, as nova-snark's CompressedSNARK::prove does); created on first use
  vdfnova::Arena arena[2];                 // compress / verify_compressed scratch, per side (freed with the set)
  vdf_ctx* aux_ctx2 = nullptr;             // ... and a third for the primary side's second opening (compress_host.cpp ipa_prove_two_queues)
  std::mutex aux_mu;                       // ... one compression at a time uses it: concurrent vdf_nova_compress calls under ONE parameter
                                           // set take turns at the arguments (calls under different sets do not meet)
  vdf_nova_tuning tune;                    // the tuning this set was made with, and that its prover runs with
  double setup_ms[7] = {0, 0, 0, 0, 0, 0, 0};
  uint64_t digit_table_bytes[2] = {0, 0};  // HBM held by each side's digit table (vdf_nova_pp_memory)
  unsigned digit_tables_skipped = 0;       // bit s: side s asked for a digit table and went without (no room, refused window)
};

struct Circuit {            // InverseMinRootCircuit<G1>, src/nova/proof.rs:57-66, + the forward trace
  uint64_t inverse_exponent = 5;
  vdfnova::St result, input;
  uint64_t t = 0;
  std::vector<Fe> trace_xy;  // (x, y) of states 0..t: trace[0] = input, trace[t] = result
  void* d_trace = nullptr;   // the same trace in HBM (vdf_nova_circuits_upload)
};
struct vdf_circuits { std::vector<Circuit> v; vdf_ctx* ctx = nullptr; };

// NovaVDFProof::Recursive = nova-snark RecursiveSNARK: running instance + witness on both sides, the last secondary
// instance unfolded, the step counter and both z_i.  Constant size in the number of steps.
struct SideState {
  vdfnova::Inst inst;        // running relaxed instance
  void* d_z = nullptr;       // [W | u | X]
  void* d_E = nullptr;
  void* d_abc[3] = {};       // A z, B z, C z of the running instance, folded along (linear in z)
  void* d_abc2[3] = {};      // the fresh instance's
  void* d_T = nullptr;
};
struct vdf_proof {
  vdf_pp* pp = nullptr;
  size_t i = 0;
  std::vector<Fe> zi[2], z0[2];
  SideState r[2];
  // fresh secondary instance (u = 1, E = 0)
  vdfnova::Inst l2;
  void* d_l2z = nullptr;
  bool l2_committed = false;
  // The NIFS of the last secondary instance (cross term, commit(w2), commit(T2)) needs nothing of the next step: prove_step
  // launches it on its way out, the next prove_step only waits for it.  NONE: not launched / results used or overwritten;
  // INFLIGHT: on the queue, results going to h_pts[RING], h_pts[RING + 1]; DONE: collected (comm_T2 in nifs2_T, A z, B z,
  // C z and T of the fresh instance still in the secondary side's scratch vectors)
  enum { NIFS2_NONE = 0, NIFS2_INFLIGHT = 1, NIFS2_DONE = 2 };
  int nifs2 = NIFS2_NONE;
  Aff nifs2_T;
  // ... and so do the early rows of the NEXT step's cross term, when that step's rounds are already in their ring slot
  // (the lookahead): launched on the way out of a step too; valid for the step that finds this slot and circuit
  bool tahead_valid = false;
  bool poisoned = false;         // a fused fold was begun and the stencil that finishes it never reached its queue (a device failure): the running
                                 // instance is half folded and every later call on this proof is refused
  int tahead_slot = -1;
  size_t tahead_k = 0;
  const vdf_circuits* tahead_circuits = nullptr;
  // fresh primary z: ring of slots, the MinRoot segment of a later step is filled (and committed) ahead of time on ctx2
  static constexpr int DEPTH = 1, RING = DEPTH + 2;
  vdf_ctx* ctx2[DEPTH] = {};
  vdf_ctx* ctx3 = nullptr;       // the early rows of a step's cross term and their commitment: a queue of their own, so that
                                 // they start with the step and not behind the previous lookahead's commitment
  void* d_z2s[RING] = {};
  void* d_traces[DEPTH] = {};
  void* d_packed[DEPTH] = {};    // scalars of the packed commitment (the reference's circuit), one buffer per lookahead queue
  int slot = 0;
  struct Ahead { vdfnova::St result, input; int slot; };
  const vdf_circuits* ahead_circuits = nullptr;
  size_t ahead_k = 0;
  std::vector<Ahead> ahead;
  Fe* h_zin = nullptr;           // pinned: the step circuit's input for the early rows of T
  vdf_jac* h_pts = nullptr;      // pinned result slots: [0, RING) segment commitments per ring slot, 4 for the batches, 1 for the early rows of T
  Fe* h_stage[2] = {};           // pinned staging of a host-synthesised witness, one per side
  // the last step's by-products, for the parity tests
  vdf_nova_step_info last;
  double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace vdfnova {
const vdf_nova_tuning& default_tuning();   // defaults + the VDF_NOVA_* environment overrides, read once
bool tuning_valid(const vdf_nova_tuning& t);
int alloc_proof_buffers(vdf_proof* p);
int finalize_l2(const vdf_proof* p);      // commits to the last secondary witness if that is still pending
std::unique_ptr<StepCircuit> make_primary_circuit(const vdf_pp* pp, const Circuit* c, bool device_rounds);
std::unique_ptr<StepCircuit> make_custom_circuit(const vdf_step_circuit* c);

// ---- wire formats (wire_host.cpp; layout in include/vdf_nova.h) --------------------------------------------
constexpr char WIRE_MAGIC_SNARK[9] = "VDFSNK03";      // compressed proof
constexpr char WIRE_MAGIC_PROOF[9] = "VDFRSK02";      // running proof (checkpoint)
inline uint8_t* wire_put_fe(uint8_t* o, const Fe& v, const Field& F) { const Fe c = from_mont(v, F); memcpy(o, c.l, 32); return o + 32; }
inline bool wire_get_fe(const uint8_t* i, const Field& F, Fe* v) {
  Fe c;
  memcpy(c.l, i, 32);
  if (geq(c.l, F.m)) return false;
  *v = to_mont(c, F);
  return true;
}
// an instance on the wire: commitment(s) as 32-byte points, then (u and) X; a strict instance has no comm_E and u = 1
inline uint8_t* put_inst(uint8_t* o, const Inst& in, const Side& sd, bool relaxed) {
  pt_compress(in.comm_W, *sd.Fb, o); o += 32;
  if (relaxed) { pt_compress(in.comm_E, *sd.Fb, o); o += 32; o = wire_put_fe(o, in.u, *sd.F); }
  for (int k = 0; k < NUM_IO; ++k) o = wire_put_fe(o, in.X[k], *sd.F);
  return o;
}
inline const uint8_t* get_inst(const uint8_t* i, Inst* in, const Side& sd, bool relaxed, bool* canonical, bool* on_curve) {
  *on_curve &= pt_decompress(i, *sd.Fb, &in->comm_W); i += 32;
  if (relaxed) {
    *on_curve &= pt_decompress(i, *sd.Fb, &in->comm_E); i += 32;
    *canonical &= wire_get_fe(i, *sd.F, &in->u); i += 32;
  } else {
    memset(&in->comm_E, 0, sizeof(Aff));
    in->u = one(*sd.F);
  }
  for (int k = 0; k < NUM_IO; ++k) { *canonical &= wire_get_fe(i, *sd.F, &in->X[k]); i += 32; }
  return i;
}
constexpr size_t INST_WIRE_RELAXED = 32 * 5, INST_WIRE_STRICT = 32 * 3;
}  // namespace vdfnova
