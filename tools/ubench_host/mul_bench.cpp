#include <chrono>
#include <cstdio>
#include "host_math.hpp"
using namespace vdfhost;
int main() {
  const Field& F = field_fq();
  Fe a = from_u64(123456789, F), b = from_u64(987654321, F);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 20000000; ++i) { a = mul(a, b, F); }
  double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / 2e7;
  printf("dependent mul: %.1f ns (%llx)\n", ns, (unsigned long long)a.l[0]);
  Fe x[8]; for (int k = 0; k < 8; ++k) x[k] = from_u64(k + 3, F);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 5000000; ++i) for (int k = 0; k < 8; ++k) x[k] = mul(x[k], b, F);
  ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / 4e7;
  printf("independent mul: %.1f ns (%llx)\n", ns, (unsigned long long)x[3].l[0]);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 20000000; ++i) { a = sqr(a, F); }
  ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / 2e7;
  printf("dependent sqr: %.1f ns (%llx)\n", ns, (unsigned long long)a.l[0]);
}
