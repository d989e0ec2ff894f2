// Shared declarations of the translation units of libvdf_nova.so (minroot_host.cpp, nova_host.cpp,
// compress_host.cpp): error plumbing, the curve / field roles of the reference, the handle structs.
#pragma once
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <memory>
#include <string>
#include <vector>
#include "../../../include/vdf_nova.h"
#include "host_math.hpp"

namespace vdfnova {
using namespace vdfhost;

int fail(int code, const std::string& msg);          // records the message for vdf_nova_last_error, returns code
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
constexpr int NUM_IO = 6;                  // X = [z_in(3), z_out(3)]
constexpr uint64_t GENS_SEED = 0x4e6f7661; // "Nova": label of the generator family
// Generators by seeded try-and-increment (include/vdf_hip.h): nobody knows their discrete logarithms, which is what
// makes the Pedersen commitments binding -- the [k_i]G family of the kernel tests would not do for a proof system.
constexpr int GENS_FAMILY = VDF_GENS_TRY_AND_INCREMENT;
constexpr int PRIMARY_FIELD = VDF_FIELD_FQ;   // S1 = pallas::Scalar, src/nova/proof.rs:29
constexpr int PRIMARY_CURVE = VDF_CURVE_PALLAS;   // G1, src/nova/proof.rs:26

struct StepRecord { Aff comm_w, comm_T; Fe r; Fe X[NUM_IO]; };


}  // namespace vdfnova


using vdfhost::Aff; using vdfhost::Fe;
using vdfnova::NUM_IO;

struct vdf_pp {
  vdf_ctx* ctx = nullptr;
  uint64_t t = 0;
  size_t num_cons = 0, num_vars = 0, ncols = 0, nnz3 = 0, num_gens = 0;
  vdf_shape* shape = nullptr;
  vdf_bases* gens = nullptr;
  uint8_t digest[32];
  // Commitment to a fresh witness over 3t + 4 instead of 4t + 4 generators (vdf_minroot_step_z_packed): new_x of round
  // j is y_j - (i_0 - 1 - j), so its generator G_{3+4j} is merged into the generator of y_j, and what remains,
  // sum_j (i_0 - 1 - j) G_{3+4j} = (i_0 - 1) S0 - S1, depends on the step's counter i_0 only.
  vdf_bases* gens_w = nullptr;
  size_t num_w = 0;
  Aff S0, S1, tS0;          // S0 = sum_j G_{3+4j}, S1 = sum_j j G_{3+4j}, tS0 = t * S0
  Aff gen_u;                // the extra generator U of the inner-product arguments: synthetic generator number num_gens
  void* d_zero = nullptr;   // num_cons zero elements (satisfiability residual)
};

struct Circuit {            // InverseMinRootCircuit<G1>, src/nova/proof.rs:57-66, + the forward trace
  uint64_t inverse_exponent = 5;
  vdfnova::St result, input;
  uint64_t t = 0;
  std::vector<Fe> trace_xy;  // (x, y) of states 0..t: trace[0] = input, trace[t] = result
  void* d_trace = nullptr;   // the same trace in HBM (vdf_nova_circuits_upload)
};
struct vdf_circuits { std::vector<Circuit> v; vdf_ctx* ctx = nullptr; };

struct vdf_proof {
  vdf_pp* pp = nullptr;
  size_t i = 0;              // steps folded so far
  Fe zi[3];                  // current z_i (starts at z0)
  Aff comm_W, comm_E;        // running relaxed instance
  Fe u, X[NUM_IO];
  void* d_z1 = nullptr;      // [W | u | X] of the running instance (W aliases the front)
  void* d_z2 = nullptr;      // [W | 1 | X] of the fresh instance: the ring slot of the current step
  // Lookahead.  The fresh witness of a later step and its commitment depend on the trace only, not on the fold
  // chain, so they are computed on further contexts of the same device while the critical path of the chain (cross
  // term -> commitment of T -> challenge -> fold) runs on the first: step j's fresh work goes to context j mod DEPTH,
  // enqueued DEPTH steps early.  DEPTH + 2 ring slots: the slot step j is written to was last read by step
  // j - DEPTH - 2, whose fold is known complete (the host has synchronised the first context since).
  static constexpr int DEPTH = 1, RING = DEPTH + 2;
  vdf_ctx* ctx2[DEPTH] = {};
  void* d_z2s[RING] = {};
  void* d_wps[RING] = {};       // the same slots' packed witnesses (3t + 4 values, what the commitment is taken over)
  // correction point of the packed commitment for the counter c_i0: (c_i0 - 1) S0 - S1; consecutive steps differ by t S0
  bool c_valid = false;
  Fe c_i0;
  vdfhost::Pt c_pt;
  void* d_traces[DEPTH] = {};   // staging for traces that are not device-resident, one per lookahead context
  int slot = 0;
  // steps [ahead_k, ahead_end) of `ahead_circuits` are in flight or landed; entry j lives in slot ahead_slot0 + (j - ahead_k)
  struct Ahead { vdfnova::St result, input; int slot; };
  const vdf_circuits* ahead_circuits = nullptr;
  size_t ahead_k = 0;
  std::vector<Ahead> ahead;     // ahead[i] describes step ahead_k + i
  void* d_E = nullptr;       // running error vector
  void* d_T = nullptr;
  void* d_abc[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // Az1,Bz1,Cz1,Az2,Bz2,Cz2
  vdf_jac* h_comm = nullptr; // pinned, device-mapped result slots: [0..RING) = commitment of W2 per ring slot, [RING] = of T
  std::vector<vdfnova::StepRecord> steps;
  double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // The O(1) instance fold of step k (two 128-bit scalar multiplications on the host) is deferred: step k+1
  // performs it while the GPU works on its commitments; everything else that reads comm_W / comm_E joins first.
  struct Deferred { bool valid = false; Aff cW0, cE0, cw, cT; uint64_t r[4]; };
  mutable Deferred pending;
  void join() const;
};

namespace vdfnova {
// r = SHAKE256(digest | U1 | u2 | comm_T) squeezed to 128 bits
Fe challenge(const vdf_pp* pp, const Aff& cW, const Aff& cE, const Fe& u, const Fe* X, const Aff& cw2, const Fe* X2,
             const Aff& cT, uint64_t r_raw[4]);
Aff fold_commitment(const Aff& a, const uint64_t r_raw[4], const Aff& b);      // a + r*b on Pallas
// The verifier's replay of the instance folds over the step records (steps must not be empty).  r_out == nullptr:
// every record's challenge must be the transcript's (false otherwise); else the challenges are written to r_out.
bool fold_replay(const vdf_pp* pp, const std::vector<StepRecord>& steps, Fe* r_out, Aff* cW, Aff* cE, Fe* u, Fe X[NUM_IO]);
int alloc_proof_buffers(vdf_proof* p);

// ---- wire formats (wire_host.cpp; layout in include/vdf_nova.h) --------------------------------------------
constexpr char WIRE_MAGIC_SNARK[9] = "VDFSNK02";      // compressed proof
constexpr char WIRE_MAGIC_PROOF[9] = "VDFRSK01";      // running proof (checkpoint)
size_t wire_chain_size(size_t num_steps);
uint8_t* wire_put_chain(uint8_t* o, const char magic[8], uint64_t t, const uint8_t digest[32], const std::vector<StepRecord>& steps);
// parses the chain, replays the folds (filling every record's challenge and the folded instance), advances *in
int wire_get_chain(const uint8_t** in, size_t* len, const char magic[8], const vdf_pp* pp, std::vector<StepRecord>* steps,
                   Aff* cW, Aff* cE, Fe* u, Fe X[NUM_IO]);
inline uint8_t* wire_put_fe(uint8_t* o, const Fe& v, const Field& F) { const Fe c = from_mont(v, F); memcpy(o, c.l, 32); return o + 32; }
inline bool wire_get_fe(const uint8_t* i, const Field& F, Fe* v) {
  Fe c;
  memcpy(c.l, i, 32);
  if (geq(c.l, F.m)) return false;
  *v = to_mont(c, F);
  return true;
}
}  // namespace vdfnova
