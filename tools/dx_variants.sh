cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do
for cfg in "0 36" "128 18" "128 24" "128 36" "0 24"; do set -- $cfg
  echo -n "dx2 tile=$1 sk=$2: "; TCAR_BF16_TILE=$1 GB_SPLITK=$2 python tools/gemm_bench.py dx2 1 50 2>&1 | grep -v Warning | tail -1
done; done
