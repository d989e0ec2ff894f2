#!/bin/bash
# AddressSanitizer + UBSan over the C++ host layer (libvdf_nova.so), CPU only: a copy of the tree under /tmp gets a
# sanitised libvdf_nova.so, and the CPU tests that reach host code without a device run against it.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-/tmp/vdf_asan}
rm -rf "$W" && mkdir -p "$W"
cp -r "$ROOT"/{vdf_amd,oracle,tests,include,examples} "$W"/
cd "$W/vdf_amd/csrc"
for f in host_math minroot_host r1cs nova_host compress_host wire_host; do
  g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -pthread -c host/$f.cpp -o "$W/$f.o"
done
g++ -shared -fPIC -fsanitize=address,undefined "$W"/*.o -o ../libvdf_nova.so -L.. -lvdf_hip -pthread -Wl,-rpath,'$ORIGIN'
cd "$W"
LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
  python -m pytest tests/test_minroot_host.py tests/test_wire.py tests/test_nova_host.py -x -q -m "not gpu"
# ThreadSanitizer over the witness synthesis (helper threads, buffer pool, early / late halves): the stand-alone driver
# tools/host_synth_bench.cpp (an instrumented library under an uninstrumented Python reports the interpreter's own threads).
# libvdf_hip.so stays out of it -- the HIP runtime does not start under TSan -- so the entry points of its ABI that the
# host objects name get stand-ins here, in /tmp only; the host-only synthesis the driver runs never reaches them.
T=$W/tsan && mkdir -p "$T" && cd "$W/vdf_amd/csrc"
for f in host_math minroot_host r1cs nova_host compress_host wire_host; do
  g++ -O1 -g -std=c++17 -fPIC -fsanitize=thread -pthread -c host/$f.cpp -o "$T/$f.o"
done
cd "$T"
nm -u *.o | grep -o "vdf_[a-z_0-9]*\|mult_pippenger[a-z_]*" | sort -u > undef.txt
nm --defined-only *.o | awk '$2 == "T" {print $3}' | sort -u > def.txt
comm -23 undef.txt def.txt | while read sym; do echo "int $sym() { return 1; }"; done > stubs.c
gcc -c stubs.c -o stubs.o
g++ -O1 -g -std=c++17 -fsanitize=thread -pthread -I "$W/include" "$ROOT/tools/host_synth_bench.cpp" host_math.o minroot_host.o r1cs.o nova_host.o \
  compress_host.o wire_host.o stubs.o -o bench
TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1" ./bench
