/* A complete client of the two C ABIs in plain C: no Python, no torch, no HIP headers.
 *
 *   forward MinRoot evaluation (host, sequential: the delay itself)      vdf_nova_eval_and_make_circuits
 *   Nova proof, one prove_step per 2^k iterations (GPU)                  vdf_nova_prove_recursively
 *   verification of the recursive proof                                  vdf_nova_verify
 *   compression and verification of the compressed proof (GPU)           vdf_nova_compress / vdf_nova_verify_compressed
 *   the compressed proof as bytes, decoded again as a verifier would     vdf_nova_snark_serialize / _deserialize
 *
 * The flow of the reference's own test (/root/reference/src/nova/proof.rs:403-451).
 * Build:  cc -O2 examples/prove_chain.c -Iinclude -Lvdf_amd -lvdf_nova -lvdf_hip -Wl,-rpath,'$ORIGIN/../vdf_amd' -o examples/prove_chain
 * Run:    examples/prove_chain [log2 iterations per step = 10] [steps = 3] [x0 = 123 | 64 hex digits] [i0 = 0] [file]
 *         x0: a small integer, or a field element as the 32 bytes of a vdf_fe in hex (Montgomery form, as the ABI passes it);
 *         file: where the compressed proof's wire bytes go.  It prints the parameters' digest and an FNV-1a-64 of those bytes,
 *         so a caller can compare the run with another client's (tests/test_gpu_c_client.py does, against vdf_amd.nova and
 *         against the committed vector tests/golden/vectors.json "wire_ivc_t2_reference").
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "vdf_nova.h"

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

#define CHECK(expr, what)                                                                                         \
  do {                                                                                                              \
    int rc_ = (expr);                                                                                               \
    if (rc_ != VDF_OK) { fprintf(stderr, "%s failed (%d): %s\n", what, rc_, vdf_nova_last_error()); return 1; }   \
  } while (0)

int main(int argc, char** argv) {
  const int log2t = argc > 1 ? atoi(argv[1]) : 10;
  const size_t steps = argc > 2 ? (size_t)atoi(argv[2]) : 3;
  if (log2t < 1 || log2t > 20 || steps < 1 || steps > 1000) { fprintf(stderr, "usage: prove_chain [log2 t] [steps]\n"); return 2; }
  const uint64_t t = 1ull << log2t;
  const char* x0_arg = argc > 3 ? argv[3] : "123";
  const uint64_t i0 = argc > 4 ? strtoull(argv[4], NULL, 10) : 0;
  const char* wire_path = argc > 5 ? argv[5] : NULL;

  int device = 0;
  vdf_ctx* ctx = NULL;
  if (vdf_ctx_create(&device, 1, &ctx) != VDF_OK) { fprintf(stderr, "no GPU: %s\n", vdf_last_error(NULL)); return 1; }

  /* initial state x = 123, y = 0, i = 0 (benches/nova.rs:24-26), as Montgomery field elements */
  vdf_state initial;
  if (strlen(x0_arg) == 64) {
    uint8_t raw[32];
    for (int k = 0; k < 32; ++k) {
      unsigned byte = 0;
      if (sscanf(x0_arg + 2 * k, "%2x", &byte) != 1) { fprintf(stderr, "x0: not hex\n"); return 2; }
      raw[k] = (uint8_t)byte;
    }
    memcpy(&initial.x, raw, 32);
  } else {
    CHECK(vdf_minroot_element(VDF_FIELD_FQ, strtoull(x0_arg, NULL, 10), &initial.x), "element");
  }
  CHECK(vdf_minroot_element(VDF_FIELD_FQ, 0, &initial.y), "element");
  CHECK(vdf_minroot_element(VDF_FIELD_FQ, i0, &initial.i), "element");

  double a = now_ms();
  vdf_pp* pp = NULL;
  CHECK(vdf_nova_public_params(ctx, t, &pp), "public_params");
  printf("public_params(2^%d): %.0f ms\n", log2t, now_ms() - a);
  uint8_t digest[32];
  CHECK(vdf_nova_pp_digest(pp, digest), "pp_digest");
  printf("digest: ");
  for (int k = 31; k >= 0; --k) printf("%02x", digest[k]);       /* (little-endian bytes, printed as the number) */
  printf("\n");

  a = now_ms();
  vdf_fe z0[3];
  vdf_circuits* circuits = NULL;
  CHECK(vdf_nova_eval_and_make_circuits(VDF_MODE_LTR_ADDCHAIN_SEQUENTIAL, t, steps, &initial, z0, &circuits), "eval_and_make_circuits");
  printf("forward evaluation of %zu x 2^%d rounds (host): %.0f ms\n", steps, log2t, now_ms() - a);
  CHECK(vdf_nova_circuits_upload(ctx, circuits), "circuits_upload");

  a = now_ms();
  vdf_proof* proof = NULL;
  CHECK(vdf_nova_prove_recursively(pp, circuits, t, z0, &proof), "prove_recursively");
  printf("prove_recursively, %zu steps: %.2f ms\n", steps, now_ms() - a);

  const vdf_fe zi[3] = {initial.x, initial.y, initial.i};
  int ok = 0;
  a = now_ms();
  CHECK(vdf_nova_verify(proof, pp, steps, z0, zi, &ok), "verify");
  printf("verify: %s (%.1f ms)\n", ok ? "true" : "FALSE", now_ms() - a);
  int all_ok = ok;

  a = now_ms();
  vdf_snark* snark = NULL;
  CHECK(vdf_nova_compress(proof, pp, &snark), "compress");
  printf("compress: %.1f ms, argument %zu bytes\n", now_ms() - a, vdf_nova_snark_size(snark));
  a = now_ms();
  CHECK(vdf_nova_verify_compressed(snark, pp, steps, z0, zi, &ok), "verify_compressed");
  printf("verify (compressed): %s (%.1f ms)\n", ok ? "true" : "FALSE", now_ms() - a);
  all_ok = all_ok && ok;

  /* what travels to a verifier: bytes; what it does with them: decode under its own parameters, verify */
  const size_t wire_len = vdf_nova_snark_serialized_size(snark);
  uint8_t* wire = (uint8_t*)malloc(wire_len);
  vdf_snark* received = NULL;
  CHECK(vdf_nova_snark_serialize(snark, wire, wire_len), "serialize");
  CHECK(vdf_nova_snark_deserialize(pp, wire, wire_len, &received), "deserialize");
  CHECK(vdf_nova_verify_compressed(received, pp, steps, z0, zi, &ok), "verify_compressed (decoded)");
  printf("compressed proof on the wire: %zu bytes; decoded and verified: %s\n", wire_len, ok ? "true" : "FALSE");
  all_ok = all_ok && ok;
  uint64_t fnv = 0xcbf29ce484222325ull;
  for (size_t k = 0; k < wire_len; ++k) { fnv ^= wire[k]; fnv *= 0x100000001b3ull; }
  printf("wire fnv1a64: %016llx\n", (unsigned long long)fnv);
  if (wire_path) {
    FILE* f = fopen(wire_path, "wb");
    if (!f || fwrite(wire, 1, wire_len, f) != wire_len) { fprintf(stderr, "cannot write %s\n", wire_path); return 1; }
    fclose(f);
  }
  free(wire);
  vdf_nova_snark_free(received);

  vdf_nova_snark_free(snark);
  vdf_nova_proof_free(proof);
  vdf_nova_circuits_free(circuits);
  vdf_nova_pp_free(pp);
  vdf_ctx_destroy(ctx);
  return all_ok ? 0 : 1;
}
