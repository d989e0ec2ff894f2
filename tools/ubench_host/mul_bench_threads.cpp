// N threads each timing their own dependent multiplication chain, five rounds: shows whether the box gives every thread a
// full core (SMT siblings, other tenants).  g++ -O3 -std=c++17 -pthread -I vdf_amd/csrc/host -I include -I vdf_amd/csrc
//   tools/ubench_host/mul_bench_threads.cpp vdf_amd/csrc/host/host_math.cpp -o /tmp/mul_threads && /tmp/mul_threads 8
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sched.h>
#include <thread>
#include <vector>
#include "host_math.hpp"
using namespace vdfhost;
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8;
  const Field& F = field_fq();
  for (int round = 0; round < 5; ++round) {
    std::vector<double> ns(n);
    std::vector<int> cpu(n);
    std::vector<std::thread> th;
    for (int t = 0; t < n; ++t)
      th.emplace_back([&, t] {
        Fe a = from_u64(123456789 + t, F), b = from_u64(987654321, F);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 3000000; ++i) a = mul(a, b, F);
        ns[t] = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / 3e6 + (a.l[0] & 1) * 1e-9;
        cpu[t] = sched_getcpu();
      });
    for (auto& x : th) x.join();
    printf("round %d:", round);
    for (int t = 0; t < n; ++t) printf("  %.1f ns (cpu %d)", ns[t], cpu[t]);
    printf("\n");
  }
}
