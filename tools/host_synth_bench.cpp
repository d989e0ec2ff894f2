#include <algorithm>
#include <chrono>
#include <vector>
#include <cstdio>
#include <cstring>
#include "vdf_nova.h"
// the point (-1, 2) lies on both Pasta curves (y^2 = x^3 + 5): real points for the in-circuit folds, which a run over
// identity inputs would skip most of
static const uint64_t MOD[2][4] = {{0x992d30ed00000001ull, 0x224698fc094cf91bull, 0, 0x4000000000000000ull},    // Fp
                                   {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0, 0x4000000000000000ull}};   // Fq
static vdf_affine base_point(int field, bool negate_y) {
  vdf_affine g;
  vdf_fe one, two;
  vdf_minroot_element(field, 1, &one);
  vdf_minroot_element(field, 2, &two);
  const uint64_t* m = MOD[field == VDF_FIELD_FQ ? 1 : 0];
  auto neg = [&](const vdf_fe& a) { vdf_fe r; unsigned __int128 br = 0; for (int i = 0; i < 4; ++i) { unsigned __int128 d = (unsigned __int128)m[i] - a.l[i] - (uint64_t)br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; } return r; };
  g.x = neg(one);
  g.y = negate_y ? neg(two) : two;
  return g;
}
int main() {
  vdf_nova_aug_inputs in; memset(&in, 0, sizeof(in));
  // i = 1 (Montgomery): use vdf_minroot_element
  vdf_minroot_element(VDF_FIELD_FQ, 1, &in.i);
  vdf_state res, inp; memset(&res,0,sizeof(res)); memset(&inp,0,sizeof(inp));
  static vdf_fe W[1<<14]; vdf_fe X[2], zn[3]; size_t nv, nc;
  for (int side = 0; side < 2; ++side) {
    if (side == 1) vdf_minroot_element(VDF_FIELD_FP, 1, &in.i);
    const int fld = side == 0 ? VDF_FIELD_FQ : VDF_FIELD_FP;       // the circuit's field holds the other curve's coordinates
    in.U_comm_W = in.U_comm_E = base_point(fld, true);
    in.u_comm_W = in.T = base_point(fld, false);
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
