#!/bin/bash
# Diagnostic library: the product objects with gemm_f32.hip recompiled so that EVERY small-GEMM launch keeps two 64-deep stages in flight
# (tcar_fixed::x3_oneshot = 100 instead of 4) -> tools/micro/libtcar_hip_x3ring.so; A/B with TCAR_LIB=<that file>.  Run after the normal build.
# `build_x3ring.sh deep`: instead, long-K launches of at most 256 workgroups take 128-deep stages (tcar_fixed::x3_deep = 1).
# `build_x3ring.sh bwdhi`: instead, the session-side backward GEMMs contract plain bf16 operands (tcar_fixed::bwd_small_hi = 1).
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import sys; sys.path.insert(0, '.'); import tcar_amd; from tcar_amd import _lib; _lib.build()"
C=session-based-news-recommendation_amd/csrc
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize"
case "$1" in
  deep)  D=-DTCAR_FIX_X3_DEEP=1; SRC="gemm_f32" ;;
  bwdhi) D=-DTCAR_FIX_BWD_SMALL_HI=1; SRC="gemm_f32 step" ;;
  dot2)  D=-DTCAR_DIAG_DOT2=1; SRC="gemm_bf16" ;;
  dot2b) D=-DTCAR_DIAG_DOT2=2; SRC="gemm_bf16" ;;
  dot2c) D=-DTCAR_DIAG_DOT2=3; SRC="gemm_bf16" ;;                  # ... with the pairs taken from the vector's elements instead of a bit cast                  # ... behind a workgroup barrier: no wave's MFMAs beside another wave's epilogue                  # the logits epilogue's rounded sum through v_dot2c_f32_bf16 (WRONG sums in the step)       # session-side backward GEMMs on plain bf16 operands (hi-only backward precision)
  *)     D=-DTCAR_FIX_X3_ONESHOT=100; SRC="gemm_f32" ;;
esac
objs=$(ls $C/*.o)
for s in $SRC; do
  /opt/rocm/bin/hipcc $F $D -c $C/$s.hip -o /tmp/${s}_diag.o
  objs=$(echo "$objs" | grep -v "/$s.o"); objs="$objs"$'\n'"/tmp/${s}_diag.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/libtcar_hip_x3ring.so $objs
ls -la tools/micro/libtcar_hip_x3ring.so
