#!/bin/bash
# Diagnostic library for tools/obs1_probe.py: the product objects with score.hip recompiled under -DTCAR_OBS1_DIAG (the LDS-staged
# first form of reduce_dact_onehot, DESIGN.md §7 observation 1) -> tools/micro/libtcar_hip_obs1.so.  Run after the normal build.
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import sys; sys.path.insert(0, '.'); import tcar_amd; from tcar_amd import _lib; _lib.build()"
C=session-based-news-recommendation_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DTCAR_OBS1_DIAG -c $C/score.hip -o /tmp/score_obs1.o
objs=$(ls $C/*.o | grep -v "/score.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/libtcar_hip_obs1.so $objs /tmp/score_obs1.o
ls -la tools/micro/libtcar_hip_obs1.so
# the same diagnostic kernel WITHOUT the SLP vectorizer (no v_pk_fma_f32 in its expansion loop): is the observation tied to the packed op?
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DTCAR_OBS1_DIAG -fno-slp-vectorize -c $C/score.hip -o /tmp/score_obs1_noslp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/libtcar_hip_obs1_noslp.so $objs /tmp/score_obs1_noslp.o
ls -la tools/micro/libtcar_hip_obs1_noslp.so
