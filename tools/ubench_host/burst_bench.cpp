// 127 dependent point doublings (one in-circuit fold's doubling chain) as ONE burst after a pause of `gap` microseconds
// spent in a pause loop or in a multiply loop: does a core that only spins between jobs run the job at full clock?
// g++ -O3 -std=c++17 -I vdf_amd/csrc/host -I include -I vdf_amd/csrc tools/ubench_host/burst_bench.cpp vdf_amd/csrc/host/host_math.cpp -o /tmp/burst
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "host_math.hpp"
using namespace vdfhost;
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const Field& F = field_fq();
  Aff g; g.x = sub(zero(), one(F), F); g.y = from_u64(2, F);
  for (int mode = 0; mode < 3; ++mode)
    for (double gap : {0.0, 100.0, 400.0, 2000.0}) {
      std::vector<double> t;
      Pt p = pt_from_aff(g, F);
      uint64_t dummy = 1;
      for (int rep = 0; rep < 300; ++rep) {
        const double until = now_us() + gap;
        while (now_us() < until) {
          if (mode == 0) __builtin_ia32_pause();
          else if (mode == 1) { for (int i = 0; i < 32; ++i) dummy = dummy * 6364136223846793005ull + 1; asm volatile("" : : "r"(dummy)); }
          else { Fe a = p.x; for (int i = 0; i < 8; ++i) a = mul(a, p.y, F); asm volatile("" : : "r"(a.l[0])); }
        }
        const double t0 = now_us();
        for (int i = 0; i < 127; ++i) p = pt_dbl(p, F);
        t.push_back(now_us() - t0);
      }
      std::sort(t.begin(), t.end());
      printf("%s gap %6.0f us: burst median %.1f us  p10 %.1f  p90 %.1f   (%llx)\n", mode == 0 ? "pause   " : mode == 1 ? "int mul " : "field mul",
             gap, t[150], t[30], t[270], (unsigned long long)p.x.l[0]);
    }
}
