// Host round trip of a tiny launch (the compressor's sum-check rounds and IPA rounds are ~70 of them in a row: launch, wait,
// hash, launch): how long from the launch call to the host SEEING the result, by the way the host waits.
//   sync   hipStreamSynchronize
//   query  spinning on hipStreamQuery
//   event  hipEventRecord + spinning on hipEventQuery
//   flag   the kernel stores a sequence number into pinned host memory after its result; the host spins on that word
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench/sync_probe.hip -o tools/ubench/sync_probe
// run:   tools/ubench/sync_probe [iterations]      (also under HSA_ENABLE_INTERRUPT=0)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_tiny(const unsigned* in, unsigned* out, volatile unsigned* flag, unsigned seq, int spin) {
  // a few microseconds of dependent work (the reduce kernels' last block is about this long)
  unsigned v = in[threadIdx.x & 63];
  for (int i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;
  if (threadIdx.x == 0) {
    out[0] = v; out[1] = seq;
    __threadfence_system();
    *flag = seq;
  }
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned *d_in, *h_out, *d_out_dev;
  CK(hipMalloc(&d_in, 256));
  CK(hipMemset(d_in, 1, 256));
  CK(hipHostMalloc((void**)&h_out, 256, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&d_out_dev, h_out, 0));
  volatile unsigned* h_flag = h_out + 16;
  unsigned* d_flag = d_out_dev + 16;
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const char* names[4] = {"sync", "query", "event", "flag"};
  for (int spin : {0, 2000}) {
    for (int mode = 0; mode < 4; ++mode) {
      std::vector<double> t(iters);
      unsigned seq = 0;
      *h_flag = 0;
      for (int it = -50; it < iters; ++it) {
        ++seq;
        const double t0 = now_us();
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d_in, d_out_dev, d_flag, seq, spin);
        if (mode == 0) CK(hipStreamSynchronize(s));
        else if (mode == 1) { while (hipStreamQuery(s) == hipErrorNotReady) {} }
        else if (mode == 2) { CK(hipEventRecord(ev, s)); while (hipEventQuery(ev) == hipErrorNotReady) {} }
        else { while (*h_flag != seq) {} }
        const double t1 = now_us();
        if (h_out[1] != seq) { fprintf(stderr, "result not visible (mode %s)\n", names[mode]); return 2; }
        if (it >= 0) t[it] = t1 - t0;
        if (mode == 3 && (it & 255) == 255) CK(hipStreamSynchronize(s));     // (keeps the runtime's bookkeeping of finished launches short)
      }
      CK(hipStreamSynchronize(s));
      std::sort(t.begin(), t.end());
      printf("spin %5d  %-5s  median %7.2f us  p10 %7.2f  p90 %7.2f  p99 %7.2f\n", spin, names[mode], t[iters / 2], t[iters / 10],
             t[iters * 9 / 10], t[iters * 99 / 100]);
    }
  }
  // back-to-back dependent pairs: launch A, wait (by mode), launch B, wait -- the sum-check's shape
  return 0;
}
