#!/bin/bash
# The whole library built with -fno-slp-vectorize (no SLP-packed v_pk_*_f32) -> tools/micro/libtcar_hip_noslp.so, for an A/B of the
# step time against the product build (TCAR_LIB=tools/micro/libtcar_hip_noslp.so python bench.py ...).
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
C=session-based-news-recommendation_amd/csrc
ID=$(python -c "import sys; sys.path.insert(0, '.'); import tcar_amd; from tcar_amd import _lib; print(_lib.source_build_id())")
mkdir -p /tmp/noslp
pids=""
for f in $C/*.hip; do
  b=$(basename $f .hip)
  extra=""; [ "$b" == "buildid" ] && extra="-DTCAR_BUILD_ID=\"$ID\""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize $extra -c $f -o /tmp/noslp/$b.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/libtcar_hip_noslp.so /tmp/noslp/*.o
ls -la tools/micro/libtcar_hip_noslp.so
