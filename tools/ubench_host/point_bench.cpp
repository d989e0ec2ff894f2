#include <chrono>
#include <cstdio>
#include "host_math.hpp"
using namespace vdfhost;
int main() {
  const Field& F = field_fq();
  Aff g; g.x = sub(zero(), one(F), F); g.y = from_u64(2, F);
  Pt p = pt_from_aff(g, F);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 1000000; ++i) p = pt_dbl(p, F);
  double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / 1e6;
  printf("pt_dbl %.0f ns (%llx)\n", ns, (unsigned long long)p.x.l[0]);
  Pt q = pt_from_aff(g, F), acc = pt_dbl(q, F);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 1000000; ++i) acc = pt_add(acc, q, F);
  ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / 1e6;
  printf("pt_add %.0f ns (%llx)\n", ns, (unsigned long long)acc.x.l[0]);
  Fe a = from_u64(3, F), b = from_u64(5, F);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 10000000; ++i) a = sub(add(a, b, F), b, F);
  ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / 2e7;
  printf("add/sub %.1f ns (%llx)\n", ns, (unsigned long long)a.l[0]);
}
