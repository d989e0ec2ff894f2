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
  void* d_z2 = nullptr;      // [W | 1 | X] of the fresh instance
  void* d_E = nullptr;       // running error vector
  void* d_T = nullptr;
  void* d_abc[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // Az1,Bz1,Cz1,Az2,Bz2,Cz2
  void* d_trace = nullptr;
  vdf_jac* h_comm = nullptr; // pinned, device-mapped result slots: [0] = commitment of W2, [1] = commitment of T
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
constexpr char WIRE_MAGIC_SNARK[9] = "VDFSNK01";      // compressed proof
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
