#!/bin/bash
# round 6, GPU session 1: the new tests (RCCL at world 1, trained-run parity at N = 46,033), ring A/B of the two gradient GEMMs
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
S="bash tools/gpu_session.sh r06s1"
$S "tests:rccl or globo_catalog_size or small_tables"
gb() { echo -n "$*: "; env GB_SUM=1 "${@:1:$#-1}" python tools/gemm_bench.py ${!#} 1 50 2>&1 | grep -v Warning | tail -1; }
for r in 1 2; do
  gb GB_TILE=0 de2; gb GB_TILE=1923 de2; gb GB_TILE=1922 de2; gb GB_TILE=1283 de2; gb GB_TILE=128 de2
  gb TCAR_BF16_KS=2 dx2; gb TCAR_BF16_KS=4 dx2; gb TCAR_BF16_KS=1 dx2
  gb GB_TILE=0 TCAR_BF16_KS=2 both2; gb GB_TILE=1923 TCAR_BF16_KS=2 both2; gb GB_TILE=1923 TCAR_BF16_KS=4 both2; gb GB_TILE=0 TCAR_BF16_KS=4 both2; gb GB_TILE=1923 TCAR_BF16_KS=1 both2
done 2>&1 | tee gpurun_out/r06s1_gemm_ab.txt
$S "ab:2:TCAR_BF16_TILE=0|TCAR_BF16_TILE=1923|TCAR_BF16_TILE=1923 TCAR_BF16_KS=4|TCAR_BF16_KS=4|TCAR_BF16_TILE=1923 TCAR_BF16_KS=1"
$S bench:default
