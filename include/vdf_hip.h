/* vdf_hip.h -- C ABI of the MI355X-native Nova/MinRoot hot path (libvdf_hip.so).
 *
 * This is the drop-in boundary (seam B2).  Each entry point
 * names the reference interface it replaces.  The reference is Rust; the third-party
 * crates it calls through (pasta-msm 0.1.1, nova-snark 0.8.0, pasta_curves 0.4.0,
 * Cargo.toml:15-18) are where these operations live today, reached from
 * `RecursiveSNARK::prove_step` at /root/reference/src/nova/proof.rs:342-349.
 *
 * Conventions
 *   - All integers little-endian.  A field element is 4 x u64 limbs of x*2^256 mod m
 *     (Montgomery form), the in-memory form of pasta_curves with `repr-c`
 *     (Cargo.toml:17).  Inputs must be canonical (< m); the hot path does not check
 *     (like pasta-msm); vdf_bases_validate checks a generator table once (VDF_ERR_NONCANONICAL).
 *   - affine = {x, y}, identity = (0, 0).  jac = {x, y, z} Jacobian, identity z = 0.
 *     Jacobian outputs are not unique; parity is defined on the affine normalisation.
 *   - Every data pointer may be a HOST pointer or a DEVICE (hipMalloc / torch) pointer;
 *     the library classifies each one with hipPointerGetAttributes.  Host buffers are
 *     staged through device memory (PCIe-inclusive); device buffers are used in place.
 *   - Caller owns every buffer.  The library keeps no caller pointer past a call; only the
 *     opaque handles (vdf_ctx, vdf_bases, vdf_shape) own device memory.
 *   - Calls are blocking unless the context is in async mode AND every buffer of the call
 *     is device-resident; then work is enqueued on the context's HIP stream.
 *   - Re-entrant across threads on one context (per-context mutex); no global mutable state
 *     except the lazily created default context of the two `mult_pippenger_*` shims and, when switched on,
 *     their generator cache (vdf_shim_set_cache).
 *   - No C++ exception or abort crosses this boundary: int status + vdf_last_error().
 *   - There is NO CPU back-end: vdf_ctx_create fails with VDF_ERR_NO_DEVICE without a GPU.
 */
#ifndef VDF_HIP_H
#define VDF_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } vdf_fe;            /* pasta_curves Fp / Fq, repr-c */
typedef struct { vdf_fe x, y; } vdf_affine;          /* pallas::Affine / vesta::Affine */
typedef struct { vdf_fe x, y, z; } vdf_jac;          /* pallas::Point / vesta::Point */

typedef struct vdf_ctx vdf_ctx;
typedef struct vdf_bases vdf_bases;
typedef struct vdf_shape vdf_shape;

enum {
  VDF_OK = 0,
  VDF_ERR_BAD_ARG = 1,
  VDF_ERR_BAD_LENGTH = 2,
  VDF_ERR_NONCANONICAL = 3,
  VDF_ERR_DEVICE = 4,
  VDF_ERR_OOM = 5,
  VDF_ERR_NO_DEVICE = 6
};

enum { VDF_CURVE_PALLAS = 0, VDF_CURVE_VESTA = 1 };   /* G1 / G2, src/nova/proof.rs:26-27 */
enum { VDF_FIELD_FP = 0, VDF_FIELD_FQ = 1 };          /* S2 = Fp, S1 = Fq, src/nova/proof.rs:29-30 */

/* ---- context ------------------------------------------------------------------------ */
/* One context drives one GPU.  device_ids/n_devices: n_devices must be 1 (one process per
 * GPU; multi-GPU jobs create one context per rank, or one per device in a single process, and
 * combine them with vdf_msm_sharded / vdf_msm_multi).  No reference counterpart (the
 * reference is CPU-only); this is the handle the FFI shim would keep in a OnceCell. */
int  vdf_ctx_create(const int* device_ids, int n_devices, vdf_ctx** out);
void vdf_ctx_destroy(vdf_ctx* ctx);
/* The budget of hardware queues.  The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (this
 * library asks for 16 when it makes the process's first HIP call and the variable is unset; the runtime's default is 4); with
 * more streams than queues, kernels of one stream wait behind another's (measured with 8: two provers and a compression -- 11
 * streams -- ran slower than one prover; with 16 they run at the two-prover rate).  A context made HERE takes
 * its stream from a per-device pool instead of opening one: a new stream while the device's streams known to this library
 * (vdf_ctx_create contexts + pooled ones) number less than the budget minus one (left to the host's own streams: torch's,
 * a caller's copies), otherwise it SHARES the least used pooled stream of its role -- VDF_QUEUE_SIDE contexts (work with a
 * step's slack: a prover's look-ahead, a compression's second and third queue) with each other before they share with a
 * VDF_QUEUE_CRITICAL one (a prover's early rows).  Sharing serialises the sharers' work and nothing else: results do not
 * change.  libvdf_nova.so makes all of its internal contexts this way, so a host may run any number of provers and
 * compressions in one process; contexts from vdf_ctx_create always own their stream (the host decides how many it opens). */
#define VDF_QUEUE_CRITICAL 1
#define VDF_QUEUE_SIDE 2
int  vdf_ctx_create_pooled(const int* device_ids, int n_devices, int role, vdf_ctx** out);
/* ... and a queue that works HAND IN HAND with `parent`'s (a prover's look-ahead and early rows beside its chain).  The
 * hardware takes queues onto its pipes in the order they were created, and which pipes a prover's three queues sit on is worth
 * 10-14 % of its rate (measured: 1,114-1,125 prove_step/s when they were created back to back, 979-1,034 with one or two
 * foreign streams opened in between; profiles/r05_single_chain_vs_padding.txt).  vdf_ctx_create therefore opens two more
 * streams right behind the context's own and keeps them; this call hands them out (VDF_QUEUE_SIDE the first, VDF_QUEUE_CRITICAL
 * the second), whatever the host has opened since.  When they are taken (a second prover on the same context), or `parent`
 * runs on a foreign stream (vdf_ctx_set_stream), the context comes from the pool as above.  The streams live as long as any
 * context that uses them. */
int  vdf_ctx_create_pooled_near(vdf_ctx* parent, int role, vdf_ctx** out);
/* *pooled = 1 for a pooled context, *sharers = contexts on its stream (1 = its own), *device_streams = streams this library
 * holds on the context's device.  Any output may be NULL. */
int  vdf_ctx_queue_info(vdf_ctx* ctx, int* pooled, int* sharers, int* device_streams);
/* Use an existing hipStream_t (e.g. torch's current stream) instead of the context's own; NULL = back to the context's own
 * (whose neighbours vdf_ctx_create_pooled_near hands out: a prover on a foreign stream takes its queues from the pool). */
int  vdf_ctx_set_stream(vdf_ctx* ctx, void* hip_stream);
void* vdf_ctx_get_stream(vdf_ctx* ctx);
/* async != 0: calls whose buffers are all device-resident return after enqueueing. */
int  vdf_ctx_set_async(vdf_ctx* ctx, int async);
int  vdf_ctx_get_async(vdf_ctx* ctx, int* async);
int  vdf_ctx_sync(vdf_ctx* ctx);
int  vdf_ctx_device(vdf_ctx* ctx);                      /* the HIP device ordinal this context drives */
const char* vdf_last_error(vdf_ctx* ctx);              /* ctx may be NULL: last create error */

/* ---- commitment generators (Pedersen bases) ------------------------------------------ */
/* Replaces nova-snark's CommitGens (built in PublicParams::setup, src/nova/proof.rs:236):
 * the generator table lives in HBM for the life of the public parameters. */
int  vdf_bases_upload(vdf_ctx* ctx, int curve, const vdf_affine* bases, size_t n, vdf_bases** out);
/* Setup-time check of a generator table (upload itself does not look at the points, like pasta-msm): every
 * coordinate a canonical residue (else VDF_ERR_NONCANONICAL) and every point the identity (0, 0) or on
 * y^2 = x^3 + 5 (else VDF_ERR_BAD_ARG); *first_bad (optional) = smallest offending index. */
int  vdf_bases_validate(vdf_ctx* ctx, const vdf_bases* bases, size_t* first_bad);
/* Synthetic generators P_i = [k_i]G, G = (-1, 2), k_i = splitmix64-derived 64-bit (see
 * oracle/pasta.py base_dlog); stands in for nova-snark's label -> hash-to-curve derivation,
 * which is implementation-defined and unpinned. */
int  vdf_bases_generate(vdf_ctx* ctx, int curve, uint64_t seed, size_t n, vdf_bases** out);
/* Same, for the index range [start, start + n): the shard a rank owns in a multi-GPU MSM. */
int  vdf_bases_generate_range(vdf_ctx* ctx, int curve, uint64_t seed, size_t start, size_t n, vdf_bases** out);
/* Generator families.  KNOWN_DLOG is the one above (it makes full-size results checkable in O(n) through
 * sum s_i P_i = [sum s_i k_i] G).  TRY_AND_INCREMENT has no known discrete logarithms, like generators derived
 * from a hash: per index a xoshiro256** stream seeded by splitmix64(seed, index) yields candidates
 * x = 256 bits mod p; the first x with x^3 + 5 a square is taken, y = the even root (SURVEY.md 8d, config 2;
 * restated in oracle/pasta.py tai_base).  Setup-time only. */
enum { VDF_GENS_KNOWN_DLOG = 0, VDF_GENS_TRY_AND_INCREMENT = 1, VDF_GENS_LABEL_SHAKE = 2 /* vdf_bases_generate_label; a family id for the proof layer */ };
int  vdf_bases_generate_family(vdf_ctx* ctx, int curve, int family, uint64_t seed, size_t start, size_t n, vdf_bases** out);
/* Generators derived from a LABEL, as nova-snark derives its CommitGens (label -> SHAKE256 -> curve points;
 * PublicParams::setup, src/nova/proof.rs:236).  nova-snark's own encoding is not in /root/reference, so
 * this one is the build's own (restated in oracle/pasta.py label_base): for index i and counter c = 0, 1, ... one SHAKE256
 * block "vdf-gens-v1" | curve u8 | len u8 | label | i LE64 | c LE32 yields 64 bytes = x mod p; the first x != 0 with
 * x^3 + 5 a square is taken, y = the even root.  Unknown discrete logarithms; reproducible from the string alone; index-
 * addressed like the other families (a rank generates its own range).  label_len <= 64. */
int  vdf_bases_generate_label(vdf_ctx* ctx, int curve, const uint8_t* label, size_t label_len, size_t start, size_t n, vdf_bases** out);
/* Build the fixed-base table  2^(window_bits*sets*j) * P_i, j = 0..tables-1, so that an MSM
 * needs only `sets` bucket sets (sets == 0: library default; sets == windows: no table).  window_bits == 0: the
 * recommended window for this many generators (16 below 2^19 generators, 17 from there on).  A table serves MSMs over
 * any sub-range of its generators; for MSMs much shorter than the table choose the window by their length. */
int  vdf_bases_precompute(vdf_ctx* ctx, vdf_bases* bases, int window_bits, int sets);
int  vdf_bases_window(const vdf_bases* bases);            /* window of the current table, 0 without one */
/* Digit table for the SMALL commitments a prover waits on (the ~10^4-term witnesses and cross-term rows of
 * RecursiveSNARK::prove_step, src/nova/proof.rs:342-349): every multiple d * 2^(window_bits * j) * P_i, d = 1 .. 2^(window_bits-1),
 * of the generators in up to 4 disjoint index ranges.  vdf_msm / vdf_msm_batch calls whose vectors all lie inside those
 * ranges (at most 2^17 scalars per call) then skip the bucket method: signed digits select table entries and the MSM is
 * a plain sum of gathered points -- two launches, ~20 dependent additions instead of a sort and ~80.  Same group element;
 * its Jacobian representative differs from the bucket method's.  HBM: count * W * 2^(window_bits-1) * 64 bytes, W =
 * ceil(256 / window_bits) (+1 when the top digit could overflow): 852 KB per generator at window_bits = 10 (0 = that).
 * ranges = 0 drops the table.  VDF_MSM_DIRECT=0 in the environment disables the path (tuning). */
int  vdf_bases_precompute_digits(vdf_ctx* ctx, vdf_bases* bases, int window_bits, int ranges, const size_t begin[], const size_t count[]);
/* HBM a digit table at window c over `generators` generators takes (64 B x windows(c) x 2^(c-1) each), 0 for a window the
 * library refuses; and what the fixed-base table of `bases` holds (0 without one).  No device work. */
size_t vdf_digit_table_bytes(int window_bits, size_t generators);
size_t vdf_bases_table_bytes(const vdf_bases* bases);
int  vdf_bases_digit_window(const vdf_bases* bases);      /* window of the digit table, 0 without one */
size_t vdf_bases_digit_table_bytes(const vdf_bases* bases); /* HBM held by the digit table, 0 without one */
int  vdf_bases_download(vdf_ctx* ctx, const vdf_bases* bases, size_t offset, size_t n, vdf_affine* out);
size_t vdf_bases_len(const vdf_bases* bases);
const void* vdf_bases_device_ptr(const vdf_bases* bases);
void vdf_bases_free(vdf_bases* bases);

/* ---- multi-scalar multiplication ------------------------------------------------------ */
/* out = sum_{i<n} scalars[i] * bases[offset + i].
 * Replaces nova-snark `Group::vartime_multiscalar_mul` -> pasta_msm::pallas/vesta
 * (the Pedersen commit of W and T inside prove_step, src/nova/proof.rs:342-349; K1/K2).
 * is_mont: scalars are in Montgomery form (as pasta-msm's `is_mont = true`). */
int  vdf_msm(vdf_ctx* ctx, const vdf_bases* bases, size_t offset, const vdf_fe* scalars, size_t n,
             int is_mont, vdf_jac* out);
/* k <= 4 MSMs over the same generator table in one pipeline: out[i] = sum_j scalars[i][j] * bases[offset[i] + j],
 * j < n[i].  The results equal k vdf_msm calls bit for bit; the batch shares every launch -- one sort, one
 * bucket-accumulation grid balanced over all entries, one latency-bound tail -- which is what a fold needs when it
 * commits to the fresh witness and to the cross term under the same generators (nova-snark commit_W / commit_T).
 * `out` holds k points, host or device. */
int  vdf_msm_batch(vdf_ctx* ctx, const vdf_bases* bases, int k, const size_t offset[], const vdf_fe* const scalars[],
                   const size_t n[], int is_mont, vdf_jac out[]);
/* The same batch with the vectors arriving one at a time -- an MSM *job*.  vdf_msm_job_push(g) starts the sort and
 * bucket accumulation of vector g on a stream of its own, ordered after the work already enqueued on the context
 * (the kernel that produces the vector) and leaving the context's stream free: a fold pushes the fresh witness,
 * enqueues the cross-term kernel, pushes T -- and T's production and sort run under the witness's ALU-bound
 * accumulation.  vdf_msm_job_finish waits for all vectors, runs the one shared bucket reduction and delivers the k
 * points (host, pinned or device memory) like vdf_msm_batch; it ends the job in every case.  Needs a fixed-base
 * table with one bucket set; scalars in device memory; one job per context at a time, and no other MSM on the
 * context while it is open.  Results equal vdf_msm / vdf_msm_batch bit for bit (as affine points). */
typedef struct vdf_msm_job vdf_msm_job;
int  vdf_msm_job_begin(vdf_ctx* ctx, const vdf_bases* bases, int k, const size_t offset[], const size_t n[], int is_mont,
                       vdf_msm_job** out);
int  vdf_msm_job_push(vdf_msm_job* job, int g, const vdf_fe* scalars);
int  vdf_msm_job_finish(vdf_msm_job* job, vdf_jac out[]);
/* Window size override for tuning (0 = automatic).  It applies to calls without a fixed-base table (a table fixes its own
 * window); a batch too wide for the requested window -- bucket sets x 2^(window - 11) sort partitions, at most 8192 --
 * runs at the largest window that fits. */
int  vdf_ctx_set_msm_window(vdf_ctx* ctx, int window_bits);
/* out = sum of n Jacobian points (the combine step of a point-chunk-sharded MSM: each GPU
 * contributes one 96-byte partial, exchanged with an RCCL all-gather). */
int  vdf_point_sum(vdf_ctx* ctx, int curve, const vdf_jac* points, size_t n, vdf_jac* out);
/* ---- one MSM across the GPUs of a node (no reference counterpart: the reference is one process) ----------
 * Point-chunk sharding: a rank owns generators [start, start + count) for good (with their fixed-base table) and gets
 * the matching slice of scalars; it contributes ONE 96-byte partial; elliptic-curve addition is no RCCL reduction
 * operator, so the exchange is an all-gather of the partials (latency-bound over xGMI) and every rank sums them itself.
 *
 * One process per GPU: the host supplies the collective.  `gather` receives device pointers -- `send` (bytes) and `recv`
 * (world * bytes, rank-major) -- and the hipStream_t the partial was produced on, on which it must order the
 * collective: RCCL `ncclAllGather(send, recv, bytes, ncclChar, comm, (hipStream_t)stream)`; returns 0 on success.
 * partial / gathered: device buffers of 1 / `world` points the caller owns (they stay valid through the collective).
 * flags: VDF_SHARDED_ALWAYS_GATHER takes the collective path for world == 1 too (rehearsal on one GPU). */
typedef int (*vdf_allgather_fn)(void* user, const void* send, void* recv, size_t bytes, void* stream);
enum { VDF_SHARDED_ALWAYS_GATHER = 1 };
int  vdf_msm_sharded(vdf_ctx* ctx, const vdf_bases* shard_bases, size_t offset, const vdf_fe* shard_scalars, size_t n, int is_mont,
                     int rank, int world, vdf_allgather_fn gather, void* user, int flags, vdf_jac* partial, vdf_jac* gathered,
                     vdf_jac* out);
/* One process driving k GPUs (k <= 64): ctxs[i] / bases[i] / scalars[i] / n[i] are device i's context, generator shard
 * and scalar slice (device or host memory); all partials are computed concurrently, land in pinned host memory and are
 * summed on ctxs[0]'s device.  Blocking. */
int  vdf_msm_multi(vdf_ctx* const ctxs[], const vdf_bases* const bases[], const size_t offsets[], const vdf_fe* const scalars[],
                   const size_t n[], int k, int is_mont, vdf_jac* out);
/* Stage timing of MSM calls (HIP events on the context's stream; for bench.py's roofline leg).
 * enable != 0 records events around the stages of every following vdf_msm.  vdf_msm_timing
 * synchronises and returns, summed over the calls since the last query:
 * ms[0] = sort (digits, histogram, scans, scatter), ms[1] = bucket accumulation kernel,
 * ms[2] = tail (fix-up, bucket reduction, final), ms[3] = whole pipeline; *calls = number of MSMs. */
int  vdf_ctx_set_timing(vdf_ctx* ctx, int enable);
int  vdf_msm_timing(vdf_ctx* ctx, float ms[4], int* calls);
/* Per-launch timing of EVERY kernel a context enqueues (bench.py's per-kernel roofline of prove_step).  While enabled,
 * each launch is bracketed by two HIP events on the stream it runs on.  vdf_ctx_kernel_events synchronises the context's
 * stream and drains the launches recorded since the last call: kernel name, the ALGORITHMIC bytes the launch is priced at
 * (96 B per (base, scalar) pair on an MSM's accumulation kernel, 96 B per folded element, 64 + 32 * vars B
 * per MinRoot round, the CSR + vector bytes of a cross term; 0 for the helper kernels of a pipeline) and start / end in
 * milliseconds since a per-device origin shared by all contexts -- so the launches of several contexts (a prover's three
 * queues) can be laid on one time line.  out == NULL: *n = launches waiting, nothing drained.  Costs two event records
 * per launch while on: measure throughput with it off. */
typedef struct { char name[24]; double bytes; double start_ms, end_ms; } vdf_kernel_event;
int  vdf_ctx_set_kernel_timing(vdf_ctx* ctx, int enable);
int  vdf_ctx_kernel_events(vdf_ctx* ctx, vdf_kernel_event* out, size_t cap, size_t* n);

/* Drop-in shims with the upstream pasta-msm 0.1.1 shape (upload-on-call, default context on device 0).  The
 * signature returns nothing and an all-zero `out` is the identity -- a valid-looking commitment -- so a call that
 * cannot compute (no device, out of memory, a failed launch) prints the reason on stderr and abort()s instead of
 * returning; a caller that wants a status code uses vdf_bases_upload + vdf_msm. */
void mult_pippenger_pallas(vdf_jac* out, const vdf_affine* points, size_t npoints, const vdf_fe* scalars, bool is_mont);
void mult_pippenger_vesta(vdf_jac* out, const vdf_affine* points, size_t npoints, const vdf_fe* scalars, bool is_mont);
/* Generator cache of the shims: keep up to `entries` (0..64; 0 = off, the default; the environment variable
 * VDF_SHIM_CACHE sets the initial value) generator arrays resident, recognised by (curve, address, length) and a
 * content hash of ALL their points (one pass over the array per call, on up to 8 host threads: ~0.3 ms at 2^19
 * points); from the second call on a set also gets its fixed-base table.  For callers whose generators do not change
 * -- nova-snark's CommitGens -- the unmodified pasta-msm call then costs a hash, a scalar upload and a table MSM instead
 * of a 64-byte-per-point upload and a table-less MSM.  An array rewritten in place, anywhere, hashes differently and is
 * uploaded again.  The address is used as a key only and never dereferenced outside the call that passes it. */
int  vdf_shim_set_cache(int entries);

/* ---- R1CS shape + sparse mat-vec ------------------------------------------------------- */
/* Replaces nova-snark R1CSShape{A,B,C} (COO triples over z = (W, u, X)) and
 * `R1CSShape::multiply_vec` (K4).  rows/cols/vals[k] describe matrix k = A, B, C; values in
 * Montgomery form.  The shape is converted to CSR with a coefficient dictionary and kept in HBM. */
int  vdf_shape_create(vdf_ctx* ctx, int field, size_t num_cons, size_t num_cols,
                      const uint32_t* const rows[3], const uint32_t* const cols[3],
                      const vdf_fe* const vals[3], const size_t nnz[3], vdf_shape** out);
void vdf_shape_free(vdf_shape* shape);
int  vdf_spmv3(vdf_ctx* ctx, const vdf_shape* shape, const vdf_fe* z, vdf_fe* Az, vdf_fe* Bz, vdf_fe* Cz);

/* ---- folding vector ops ---------------------------------------------------------------- */
/* T = Az1 o Bz2 + Az2 o Bz1 - u1*Cz2 - Cz1   (nova-snark `commit_T`, K5; u2 = 1). */
int  vdf_cross_term(vdf_ctx* ctx, int field, const vdf_fe* Az1, const vdf_fe* Bz1, const vdf_fe* Cz1,
                    const vdf_fe* Az2, const vdf_fe* Bz2, const vdf_fe* Cz2, const vdf_fe* u1,
                    size_t n, vdf_fe* T);
/* is_zero = 1 iff all n elements are zero, decided on the device (a verifier's residual A z o B z - u C z - E is checked
 * where it lies instead of being copied out); synchronises. */
int  vdf_vec_is_zero(vdf_ctx* ctx, const vdf_fe* v, size_t n, int* is_zero);
/* out = a + r*b   (nova-snark RelaxedR1CSWitness::fold: W1 + r*W2, E1 + r*T; K6).
 * `r` is one field element (host or device).  out may alias a. */
int  vdf_axpy(vdf_ctx* ctx, int field, const vdf_fe* a, const vdf_fe* r, const vdf_fe* b, size_t n, vdf_fe* out);

/* ---- MinRoot step-circuit witness ------------------------------------------------------- */
/* Fills the 4t+1 auxiliary values `InverseMinRootCircuit::synthesize` allocates
 * (src/nova/proof.rs:107-126), in allocation order per round new_x, tmp1, tmp2, new_y
 * (src/nova/proof.rs:167, 176, 178, 181) then final_i (:122), from the forward trace
 * trace_xy[k] = (x_k, y_k), k = 0..t (trace_xy[t] is the step's `result`, trace_xy[0] its
 * `input`); i0 = the `i` of trace_xy[0].  Round-parallel. */
int  vdf_minroot_witness(vdf_ctx* ctx, int field, const vdf_fe* trace_xy, const vdf_fe* i0, uint64_t t,
                         vdf_fe* W_segment);

/* ---- fused step operations (one kernel launch each) --------------------------------------- */
/* The three entry points below do what a sequence of the calls above does, in a single launch and with
 * every single-field-element operand read from HOST memory and passed as a kernel argument: no staging
 * copy, no stream synchronisation.  They exist because one NIFS fold at 2^16 iterations is ~1 ms of GPU
 * work, so a dozen launches and half a dozen 32-byte uploads per step are a measurable share of it.
 * Vector operands must already be in device memory (VDF_ERR_BAD_ARG otherwise); results are identical,
 * bit for bit, to the unfused calls. */

/* The whole fresh column vector z = (W, u, X) of the exposed-IO MinRoot step circuit:
 *   z = [ z_in[0..3) | new_x, tmp1, tmp2, new_y per round (4t, as vdf_minroot_witness) | i0 | u | X[0..6) ]
 * i.e. vdf_minroot_witness plus the seven scalar stores around it.  z has 3 + 4t + 1 + 1 + 6 elements.
 * z_in, i0, u, X: host memory. */
int  vdf_minroot_step_z(vdf_ctx* ctx, int field, const vdf_fe* trace_xy, uint64_t t, const vdf_fe z_in[3],
                        const vdf_fe* i0, const vdf_fe* u, const vdf_fe X[6], vdf_fe* z);
/* The same, and also the witness without its new_x values:
 *   w_packed = [ z_in[0..3) | tmp1, tmp2, new_y per round (3t) | i0 ],  3t + 4 elements.
 * new_x of round j equals y_j - (i_j - 1), where y_j is z_in[1] (j = 0) or the previous round's new_y and i_j = z_in[2] - j
 * (src/nova/proof.rs:162-173): an affine image of another witness value.  A Pedersen commitment to W therefore needs
 * no term for it -- sum_j new_x_j G_j folds into the generators of the y_j and a point that depends on z_in[2] only --
 * and is an MSM over 3t + 4 instead of 4t + 4 points, with the same value (libvdf_nova.so uses this). */
int  vdf_minroot_step_z_packed(vdf_ctx* ctx, int field, const vdf_fe* trace_xy, uint64_t t, const vdf_fe z_in[3],
                               const vdf_fe* i0, const vdf_fe* u, const vdf_fe X[6], vdf_fe* z, vdf_fe* w_packed);
/* Only the variables `InverseMinRootCircuit::synthesize` allocates (src/nova/proof.rs:107-126), for a step circuit that
 * sits inside an augmented circuit: vars_per_round = 4: new_x, tmp1, tmp2, new_y per round (the reference's allocation,
 * as vdf_minroot_witness) ; vars_per_round = 3: tmp1, tmp2, new_y (the bound form, where new_x is the linear
 * combination y - i + 1 and no variable); then final_i = i0.  out has vars_per_round * t + 1 elements.  i0: host memory. */
int  vdf_minroot_step_segment(vdf_ctx* ctx, int field, const vdf_fe* trace_xy, uint64_t t, const vdf_fe* i0, int vars_per_round,
                              vdf_fe* out);
/* The reference's allocation (vars_per_round = 4) and, in the same pass, the scalars of its commitment WITHOUT the new_x
 * terms:   packed = [ tmp1, tmp2, new_y per round (3t) | final_i | y_0 | i_in | 1 ],  3t + 4 elements,
 * y_0 the y the first round reads and i_in the i it reads (= i0 + t; host memory, like i0).  new_x of round j is
 * y_j - (i_in - (j + 1)) with y_j = new_y of round j - 1 (src/nova/proof.rs:162-173), so over the generators G of `out`
 *   sum_j new_x_j G[4j]  =  y_0 G[0] + sum_(j>=1) new_y_(j-1) G[4j] - i_in S1 + S2,   S1 = sum_j G[4j],  S2 = sum_j (j + 1) G[4j]
 * and the commitment to `out` equals the MSM of `packed` over the derived generators
 *   [ G[4j+1], G[4j+2], G[4j+3] + G[4j+4] (last round: G[4t-1]) per round | G[4t] | G[0] | -S1 | S2 ]
 * -- the same group element from 3t + 4 instead of 4t + 1 terms (libvdf_nova.so commits the reference's circuit this way). */
int  vdf_minroot_step_segment_packed(vdf_ctx* ctx, int field, const vdf_fe* trace_xy, uint64_t t, const vdf_fe* i0, const vdf_fe* i_in,
                                     vdf_fe* out, vdf_fe* packed);
/* vdf_spmv3(shape, z2) followed by vdf_cross_term(Az1, Bz1, Cz1, Az2, Bz2, Cz2, u1): writes Az2, Bz2, Cz2
 * (num_cons each) and T.  u1: host memory.  (nova-snark NIFS::prove -> commit_T, K4 + K5.) */
int  vdf_nifs_cross_term(vdf_ctx* ctx, const vdf_shape* shape, const vdf_fe* z2, const vdf_fe* Az1, const vdf_fe* Bz1,
                         const vdf_fe* Cz1, const vdf_fe* u1, vdf_fe* Az2, vdf_fe* Bz2, vdf_fe* Cz2, vdf_fe* T);
/* The same over part of the rows: VDF_ROWS_INSIDE = the rows [row_begin, row_begin + row_count) only, VDF_ROWS_OUTSIDE =
 * every row but those.  The vectors are full-length either way (num_cons); only the selected rows are read and written.
 * A prover that knows part of the fresh witness early (the step circuit's own variables, made ahead of the step) runs
 * the rows that read nothing else on another context before the rest of the witness exists; the range must not hold a
 * row of more than 8 entries (those are summed by a wavefront in the OUTSIDE / ALL call). */
#define VDF_ROWS_ALL 0
#define VDF_ROWS_INSIDE 1
#define VDF_ROWS_OUTSIDE 2
int  vdf_nifs_cross_term_rows(vdf_ctx* ctx, const vdf_shape* shape, size_t row_begin, size_t row_count, int part, const vdf_fe* z2,
                              const vdf_fe* Az1, const vdf_fe* Bz1, const vdf_fe* Cz1, const vdf_fe* u1, vdf_fe* Az2, vdf_fe* Bz2,
                              vdf_fe* Cz2, vdf_fe* T);
/* The same rows for the built-in MinRoot step circuits WITHOUT the sparse matrices: the 3t + 1 constraints of
 * InverseMinRootCircuit::synthesize (src/nova/proof.rs:107-133, :219-227) are a fixed stencil over the rounds' own
 * variables -- A z2, B z2, C z2 of a row are copies of witness values and one four-term sum -- so rows
 * [row_begin, row_begin + 3t + 1) are computed from coalesced streams only (no row pointers, columns, coefficients or
 * gathers).  seg_begin: index of the first round variable in z2 (the step circuit's input x, y, i occupies the three
 * variables before it); vars_per_round: 4 = the reference's rounds (new_x, tmp1, tmp2, new_y), 3 = the bound form;
 * one_col: the constant's column (num_vars).  Exact for any z2.  The caller is responsible for the rows being that
 * stencil (libvdf_nova.so compares it with the shape's triples once, at public_params).  Vectors: device memory,
 * full length (num_cons / num_cols); u1: host memory. */
int  vdf_nifs_cross_term_minroot(vdf_ctx* ctx, int field, int vars_per_round, uint64_t t, size_t seg_begin, size_t one_col, size_t row_begin,
                                 const vdf_fe* z2, const vdf_fe* Az1, const vdf_fe* Bz1, const vdf_fe* Cz1, const vdf_fe* u1, vdf_fe* Az2,
                                 vdf_fe* Bz2, vdf_fe* Cz2, vdf_fe* T);
/* The same rows with the PREVIOUS fold of those rows applied on the way.  A prover that keeps A z, B z, C z of the running
 * instance folds them after every step (X1 <- X1 + r X2); for the stencil rows the fresh vectors X2 of the previous step are
 * exactly what this call is about to overwrite in Az2 / Bz2 / Cz2.  So, per row: Az1 += r Az2, Bz1 += r Bz2, Cz1 += r Cz2
 * (and E1 += r T_prev when E1 is not NULL; T_prev may be T itself) with the vectors' CURRENT contents, stored in place; then
 * Az2, Bz2, Cz2 and T of z2 as vdf_nifs_cross_term_minroot computes them, crossed with the FOLDED running rows and u1 (the
 * caller passes the folded u).  One pass over the rows instead of a fold over whole vectors in front of it; the caller folds
 * z, and every vector outside [row_begin, row_begin + 3t + 1), with vdf_fold_many.  r, u1: host memory. */
int  vdf_nifs_cross_term_minroot_fold(vdf_ctx* ctx, int field, int vars_per_round, uint64_t t, size_t seg_begin, size_t one_col,
                                      size_t row_begin, const vdf_fe* z2, const vdf_fe* r, vdf_fe* Az1, vdf_fe* Bz1, vdf_fe* Cz1,
                                      vdf_fe* E1, const vdf_fe* T_prev, const vdf_fe* u1, vdf_fe* Az2, vdf_fe* Bz2, vdf_fe* Cz2,
                                      vdf_fe* T);
/* acc[i] <- acc[i] + r * add[i], i < k <= 8, n[i] elements each: k vdf_axpy calls with a common r (host
 * memory).  The fold of a relaxed witness is k = 2 (W, E); a prover that keeps A z, B z, C z of the running
 * instance folds them too (they are linear in z), k = 5, instead of recomputing three sparse products. */
int  vdf_fold_many(vdf_ctx* ctx, int field, const vdf_fe* r, int k, vdf_fe* const acc[], const vdf_fe* const add[],
                   const size_t n[]);
/* Stream order across contexts of one device: work enqueued on `ctx` after this call starts only after
 * everything enqueued on `other` so far has finished (an event; the host does not wait). */
int  vdf_ctx_wait(vdf_ctx* ctx, vdf_ctx* other);
/* Marks: vdf_ctx_mark remembers the current end of the context's queue under `slot` (0 .. VDF_MARK_SLOTS-1: 0..3 are the
 * caller's on ANY context; 4..15 are used by libvdf_nova.so on the contexts it is given -- prove_step 4..7, compress 8 --
 * and never 0..3 there); vdf_ctx_sync_mark
 * blocks the host until everything enqueued before that mark has finished, while later work keeps running.  A
 * prover that looks one step ahead waits for this step's commitment without waiting for the next step's. */
#define VDF_MARK_SLOTS 16
int  vdf_ctx_mark(vdf_ctx* ctx, int slot);
int  vdf_ctx_sync_mark(vdf_ctx* ctx, int slot);
/* ... and vdf_ctx_wait_mark makes work enqueued on `ctx` from now on start only after `other`'s mark `slot` has been
 * reached -- not after whatever `other` was given since (vdf_ctx_wait would wait for that too). */
int  vdf_ctx_wait_mark(vdf_ctx* ctx, vdf_ctx* other, int slot);
/* One-shot scheduling hint for the NEXT bucket-method MSM enqueued on `ctx`: its sort runs as usual, its bucket
 * accumulation -- the one kernel of the pipeline that fills every SIMD -- starts only after `other`'s mark `slot` has
 * been reached.  A prover that commits ahead on a side queue keeps that accumulation out of the way of a short,
 * latency-critical kernel on its main queue (measured: a 10^4-term direct sum takes 190 us beside an accumulation and
 * 110 us alone).  Results are unchanged; an MSM that takes the direct-sum path ignores the hint. */
int  vdf_ctx_gate_accumulate(vdf_ctx* ctx, vdf_ctx* other, int slot);
/* Wave priority (0..3, default 3) of the latency-bound kernels of this context's bucket-method MSMs -- the sort and the
 * bucket reduction; the bucket accumulation always runs at 0 and a direct sum at 2.  A prover that commits on several
 * queues gives the queue whose result it needs LAST a lower priority than the one on its longest dependent path
 * (libvdf_nova.so: the lookahead's MSM at 1, the early rows' at 3).  Scheduling only: results are unchanged. */
int  vdf_ctx_set_light_priority(vdf_ctx* ctx, int priority);

/* ---- compression SNARK building blocks ------------------------------------------------------ */
/* The passes behind `NovaVDFProof::compress` and the verification of a compressed proof
 * (/root/reference/src/nova/proof.rs:360-368, :383 -> nova-snark 0.8.0 CompressedSNARK: sum-checks over the relaxed
 * R1CS and inner-product-argument openings).  Protocol and padding are restated in
 * oracle/spartan.py.  Vectors live in device memory, single field elements in host memory (passed as kernel
 * arguments); multilinear tables are MSB-first: index i = sum x_j 2^(k-j), binding a variable folds the upper half
 * of a table onto the lower half. */
/* out[i] = prod_j (bit_j(i) ? hi[j] : lo[j]), i < 2^k, k <= 24: eq(r, .) with lo = 1 - r, hi = r; the inner-product
 * argument's generator coefficients with lo = x^-1, hi = x. */
int  vdf_pair_table(vdf_ctx* ctx, int field, const vdf_fe* lo, const vdf_fe* hi, int k, vdf_fe* out);
/* out[i] = (prod_j (bit_j(i >> log_m) ? hi[j] : lo[j])) * pattern[i mod 2^log_m], i < 2^(k + log_m), log_m <= 4: the table over
 * the top k index bits times a vector of 2^log_m values over the low bits (host memory) -- the generator coefficients of an
 * inner-product argument that stopped at a vector of 2^log_m elements instead of halving down to one. */
int  vdf_pair_table_pattern(vdf_ctx* ctx, int field, const vdf_fe* lo, const vdf_fe* hi, int k, const vdf_fe* pattern, int log_m,
                            vdf_fe* out);
/* v[t][i] <- c_lo[t] * v[t][i] + c_hi[t] * v[t][i + n/2], i < n/2, for t < k <= 8 vectors of length n (power of two)
 * in one launch: a sum-check binding (1 - r, r); the argument's folds (x, x^-1) / (x^-1, x). */
int  vdf_fold_halves(vdf_ctx* ctx, int field, int k, vdf_fe* const v[], const vdf_fe c_lo[], const vdf_fe c_hi[], size_t n);
/* Sums over a vector pair / the five tables of the R1CS sum-check; `out` host, pinned or device:
 *   kind 0  out[0] = sum_i a[i] b[i], i < n                                    tables = {a, b}
 *   kind 1  g(0), g(2) of g(t) = sum_{i < n/2} p_t[i] q_t[i]                    tables = {p, q}
 *   kind 2  g(0), g(2), g(3) of g(t) = sum eq_t (a_t b_t - u c_t - e_t)        tables = {eq, a, b, c, e}, u: host
 *   kind 3  sum a[i] b[n/2 + i],  sum a[n/2 + i] b[i]                          tables = {a, b}
 * with f_t[i] = f[i] + t (f[n/2 + i] - f[i]). */
enum { VDF_REDUCE_DOT = 0, VDF_REDUCE_QUADRATIC_ROUND = 1, VDF_REDUCE_R1CS_ROUND = 2, VDF_REDUCE_IPA_CROSS = 3 };
int  vdf_reduce(vdf_ctx* ctx, int field, int kind, const vdf_fe* const tables[], const vdf_fe* u, size_t n, vdf_fe* out);
/* out[y] = sum_x eq[x] (A + rho B + rho^2 C)[x, y] for the num_cols columns of the shape (eq: num_cons elements). */
int  vdf_spmv3_t(vdf_ctx* ctx, const vdf_shape* shape, const vdf_fe* eq, const vdf_fe* rho, vdf_fe* out);
/* One round of the inner-product argument without materialising folded generators: with n_j the current length of
 * a, s[t] the coefficient of original generator t in its folded generator, h = n_j / 2 and r = t mod n_j,
 *   sL[t] = r >= h ? s[t] a[r - h] : 0,   sR[t] = r < h ? s[t] a[r + h] : 0,   t < n,
 * so that L = MSM(sL) and R = MSM(sR) over the original generators. */
int  vdf_ipa_scalars(vdf_ctx* ctx, int field, const vdf_fe* a, const vdf_fe* s, size_t n, size_t nj, vdf_fe* sL, vdf_fe* sR);
/* s[t] *= (t mod n_j) >= n_j / 2 ? x_hi : x_lo, t < n. */
int  vdf_scale_pattern(vdf_ctx* ctx, int field, vdf_fe* s, size_t n, size_t nj, const vdf_fe* x_lo, const vdf_fe* x_hi);

/* ---- utilities the host layer and the tests need ---------------------------------------- */
/* Element-wise Montgomery product / conversions, n elements (test + host plumbing). */
int  vdf_fe_mul(vdf_ctx* ctx, int field, const vdf_fe* a, const vdf_fe* b, size_t n, vdf_fe* out);
int  vdf_fe_to_mont(vdf_ctx* ctx, int field, const vdf_fe* a, size_t n, vdf_fe* out);
int  vdf_fe_from_mont(vdf_ctx* ctx, int field, const vdf_fe* a, size_t n, vdf_fe* out);
/* Throughput probe: each of n lanes runs `iters` dependent Montgomery multiplications
 * (roofline / issue-rate calibration for DESIGN.md; not on the prove path). */
int  vdf_fe_mul_chain(vdf_ctx* ctx, int field, const vdf_fe* a, size_t n, int iters, vdf_fe* out);
/* Process-wide tuning of the kernels (every field has a measured default; DESIGN.md says what each was worth).  The
 * environment variables of earlier rounds (VDF_MSM_*, VDF_NIFS_LANES, VDF_SHIM_CACHE) are read ONCE, the first time the
 * library needs a value, as overrides of these defaults; a host sets them here instead.  Values are read at launch time.
 * Thread-safe: a set publishes a new immutable snapshot; a call running on another thread meanwhile sees the old values or
 * the new ones at each of its launches, never a mixture within one read (set before a call for a defined outcome). */
typedef struct vdf_hip_tuning {
  uint32_t struct_size;        /* sizeof(vdf_hip_tuning) as the caller compiled it */
  int32_t msm_direct;          /* 1: MSMs inside a digit table's ranges are direct sums (msm_direct.hip); 0: bucket method only */
  int32_t direct_priority;     /* wave priority of the direct sum, 0..3 (2) */
  int32_t direct_fused;        /* 1: the direct sum's last workgroup adds the workgroup points; 0: a second launch does */
  int32_t light_priority;      /* ceiling of the wave priority of sort / fix-up / bucket-reduction kernels, 0..3 (3);
                                  a context lowers its own with vdf_ctx_set_light_priority */
  int32_t accumulate_fill;     /* resident bucket-accumulation workgroups per CU one launch is sized for, 1..3; 0 (the default) =
                                  automatic: three for a single MSM of 2^22 (scalar, window) entries or more, two below and for batches; a context overrides it with
                                  vdf_ctx_set_accumulate_fill */
  int32_t accumulate_lds;      /* bytes of unused LDS per accumulation workgroup: caps its occupancy per CU (0) */
  int32_t slice_len;           /* fixed slice length of the bucket accumulation, 1..65536; 0 = from the entry count */
  int32_t part_bits;           /* high bucket bits of the sort's first pass; -1 = about one partition per CU */
  int32_t reduction;           /* 1: bucket reduction as a matrix (k_red_sums / weights / combine); 0: segments */
  int32_t reduction_quads;     /* quads of the segment reduction, 64..65536; 0 = 8192 */
  int32_t heavy_min;           /* a bucket spanning more slices than max(this, 2 x average + 4) goes to a wavefront; 0 = 6 */
  int32_t giant_span;          /* ... and more than this to several wavefronts; 0 = 64 */
  int32_t nifs_lanes;          /* lanes per row of the fused cross term: 1, 4, 8; 0 = 8 up to 2^15 rows, else 1 */
  int32_t shim_cache;          /* generator arrays the mult_pippenger shims keep resident, 0..64 (0; vdf_shim_set_cache) */
  int32_t nifs_fused;          /* 1: rows of more than 8 entries are summed inside the cross term's launch (k_nifs_cross_f) when it
                                  runs eight lanes per row; 0: by a launch of their own before it (1) */
  int32_t fold_u128;           /* 1: vdf_fold_many with a scalar below 2^128 (every NIFS fold challenge) multiplies by its plain value
                                  without a Montgomery reduction (fe_mul_u128); 0: the general multiplication (1) */
  int32_t fixup_serial;        /* 1: the slice heads of a bucket are added by ONE lane (k_fixup_serial: 2.4 x fewer instructions per
                                  addition, a longer dependent chain; 1); 0: by a quad of lanes (k_fixup) */
  int32_t sort_staged;         /* 1: the sort's second pass lays a tile of entries out bucket by bucket in LDS and writes whole
                                  runs (k_fine_staged); 0: every entry written on its own (k_fine) */
  int32_t glv;                 /* 1: a table-less MSM over a whole generator set of 2^12 points or more uses the curve's endomorphism
                                  (2n points, 129-bit half-scalars: half the Horner chain); 0: never */
} vdf_hip_tuning;
int  vdf_hip_tuning_get(vdf_hip_tuning* out);            /* the values in force (struct_size filled in) */
int  vdf_hip_tuning_set(const vdf_hip_tuning* in);       /* VDF_ERR_BAD_ARG (nothing changed) if a field is out of range */
/* Per context: how many resident accumulation workgroups per CU its bucket-method MSMs fill (1..3; 0 = the process-wide
 * value).  A prover's side queues fill all three (their launches share the device with other queues' kernels, and a
 * grid sized for two is packed three-and-one by the dispatcher); independent MSMs in flight do better with two. */
int  vdf_ctx_set_accumulate_fill(vdf_ctx* ctx, int workgroups_per_cu);
/* Box fingerprint: every SIMD runs `iters` dependent Montgomery products (two wavefronts per SIMD, ~0.85 us per
 * iteration: 6000 iterations = 5 ms); *shader_mhz = shader clocks / 100 MHz reference ticks summed over the wavefronts
 * (the clock the device sustained under the MSM's kind of load), *kernel_ms = the launch's duration (HIP events).
 * Synchronises the context's stream.  Either output may be NULL. */
int  vdf_ctx_clock_probe(vdf_ctx* ctx, int iters, double* shader_mhz, double* kernel_ms);
/* Device memory helpers so a non-torch host (the C++ Nova layer) can keep state resident. */
int  vdf_dev_alloc(vdf_ctx* ctx, size_t bytes, void** out);
int  vdf_dev_free(vdf_ctx* ctx, void* p);
int  vdf_dev_mem_info(vdf_ctx* ctx, size_t* free_bytes, size_t* total_bytes);   /* HBM of the context's device; either may be NULL */
int  vdf_dev_memcpy(vdf_ctx* ctx, void* dst, const void* src, size_t bytes);   /* any direction */
/* Pinned host memory mapped into the device's address space.  Calls treat such a pointer like device memory
 * (used in place, no staging, no implicit synchronisation): a kernel's small result -- an MSM's point -- lands
 * in host memory by itself and is readable after vdf_ctx_sync, with no device-to-host copy. */
int  vdf_host_alloc(vdf_ctx* ctx, size_t bytes, void** out);
int  vdf_host_free(vdf_ctx* ctx, void* p);
int  vdf_dev_memset(vdf_ctx* ctx, void* dst, int value, size_t bytes);
/* Library build identification ("vdf_hip gfx950 <date>"). */
const char* vdf_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VDF_HIP_H */
