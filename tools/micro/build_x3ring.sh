#!/bin/bash
# Diagnostic library: the product objects with gemm_f32.hip recompiled so that EVERY small-GEMM launch keeps two 64-deep stages in flight
# (tcar_fixed::x3_oneshot = 100 instead of 4) -> tools/micro/libtcar_hip_x3ring.so; A/B with TCAR_LIB=<that file>.  Run after the normal build.
# `build_x3ring.sh deep`: instead, long-K launches of at most 256 workgroups take 128-deep stages (tcar_fixed::x3_deep = 1).
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import sys; sys.path.insert(0, '.'); import tcar_amd; from tcar_amd import _lib; _lib.build()"
C=session-based-news-recommendation_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize $([ "$1" == deep ] && echo -DTCAR_FIX_X3_DEEP=1 || echo -DTCAR_FIX_X3_ONESHOT=100) -c $C/gemm_f32.hip -o /tmp/gemm_f32_ring.o
objs=$(ls $C/*.o | grep -v "/gemm_f32.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/libtcar_hip_x3ring.so $objs /tmp/gemm_f32_ring.o
ls -la tools/micro/libtcar_hip_x3ring.so
