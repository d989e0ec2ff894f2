#include <chrono>
#include <cstdio>
#include <cstring>
#include "vdf_nova.h"
int main() {
  vdf_fe xs[3], out;
  for (int k = 0; k < 3; ++k) vdf_minroot_element(VDF_FIELD_FQ, 5 + k, &xs[k]);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 20000; ++i) { vdf_nova_ro_hash(VDF_FIELD_FQ, 1, xs, 3, &out); xs[0] = out; }
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20000;
  printf("native permutation (one absorb of 3): %.2f us\n", us);
}
