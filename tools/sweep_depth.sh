set -e
cd $GRAFT_REPO_ROOT
for d in 1 2 3 4 3 1; do
  sed -i "s/static constexpr int DEPTH = [0-9]*, RING = DEPTH + 2;/static constexpr int DEPTH = $d, RING = DEPTH + 2;/" vdf_amd/csrc/host/nova_internal.hpp
  (cd vdf_amd/csrc && make >/dev/null 2>&1)
  echo "== DEPTH=$d"; timeout -k 10 300 python tools/gpu_prove_time.py 16 32 2>&1 | tail -2 | head -1
done
