#include <algorithm>
#include <chrono>
#include <vector>
#include <cstdio>
#include <cstring>
#include "vdf_nova.h"
int main() {
  vdf_nova_aug_inputs in; memset(&in, 0, sizeof(in));
  // i = 1 (Montgomery): use vdf_minroot_element
  vdf_minroot_element(VDF_FIELD_FQ, 1, &in.i);
  vdf_state res, inp; memset(&res,0,sizeof(res)); memset(&inp,0,sizeof(inp));
  static vdf_fe W[1<<14]; vdf_fe X[2], zn[3]; size_t nv, nc;
  for (int side = 0; side < 2; ++side) {
    if (side == 1) vdf_minroot_element(VDF_FIELD_FP, 1, &in.i);
    vdf_nova_aug_synthesize(side, 5, 0, &in, &res, &inp, W, 1<<14, &nv, &nc, X, zn);
    std::vector<double> ms;
    for (int k = 0; k < 400; ++k) {
      auto t0 = std::chrono::steady_clock::now();
      vdf_nova_aug_synthesize(side, 5, 0, &in, &res, &inp, W, 1<<14, &nv, &nc, X, zn);
      ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(ms.begin(), ms.end());
    printf("side %d: min %.3f  median %.3f  p90 %.3f  max %.3f ms (nv %zu nc %zu)\n", side, ms[0], ms[200], ms[360], ms[399], nv, nc);
  }
}
