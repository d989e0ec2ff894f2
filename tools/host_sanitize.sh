#!/bin/bash
# AddressSanitizer + UBSan over the C++ host layer (libvdf_nova.so), CPU only: a copy of the tree under /tmp gets a
# sanitised libvdf_nova.so, and the CPU tests that reach host code without a device run against it.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-/tmp/vdf_asan}
rm -rf "$W" && mkdir -p "$W"
cp -r "$ROOT"/{vdf_amd,oracle,tests,include,examples} "$W"/
cd "$W/vdf_amd/csrc"
for f in host_math minroot_host nova_host compress_host wire_host; do
  g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -pthread -c host/$f.cpp -o "$W/$f.o"
done
g++ -shared -fPIC -fsanitize=address,undefined "$W"/*.o -o ../libvdf_nova.so -L.. -lvdf_hip -pthread -Wl,-rpath,'$ORIGIN'
cd "$W"
LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
  python -m pytest tests/test_minroot_host.py tests/test_wire.py -x -q -m "not gpu"
