#!/bin/bash
# Build a second copy of libsefd_hip.so from the sources of a git revision (default HEAD) with selected files replaced: same-box A/B runs
#   [AB_FLAGS=-DX=1] tools/ab_build.sh <out.so> [rev | WORK = the working tree] [file@rev2 ...]        then   SEFD_LIB_PATH=$PWD/<out.so> python bench.py ...
# e.g. tools/ab_build.sh gpurun_ab/base.so HEAD cgemm256.hip@cb96ec7   (everything from HEAD, that one kernel file from the older commit)
set -e
OUT=$1; REV=${2:-HEAD}; shift; shift || true
PKG=dnn-based-speech-enhancement-in-the-frequency-domain_amd
T=$(mktemp -d)
if [ "$REV" = WORK ]; then mkdir -p $T/$PKG && cp -r $PKG/csrc $T/$PKG/ && cp -r include $T/ && rm -f $T/$PKG/csrc/*.o; else git archive $REV $PKG/csrc include | tar -x -C $T; fi
for ov in "$@"; do f=${ov%@*}; r=${ov#*@}; git show $r:$PKG/csrc/$f > $T/$PKG/csrc/$f; done
cd $T/$PKG/csrc
objs=""
for s in *.hip plan.cpp; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $AB_FLAGS -c -o $s.o $s & objs="$objs $s.o"; done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $OLDPWD/$OUT $objs 2>/dev/null || /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $OUT $objs
cd - > /dev/null; rm -rf $T; ls -la $OUT
