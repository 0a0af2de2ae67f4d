#!/bin/bash
# A variant build of the diagnostic library for A/B runs (scripts/ab_chain.py, ab_libs.py): same sources, extra -D flags,
# only the named sources recompiled.  usage: bash scripts/build_variant.sh <name> "<-D flags>" <source.hip> [...]
# -> comms_rs_amd/lib/libcomms_hip_<name>.so   (MAKEVARS="FLAGS_fir_decim.hip=" in the environment overrides a make variable)
set -e
NAME=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/../comms_rs_amd/csrc"
make -s -j8 diag 2>&1 | grep -v load-store-opt || true
rm -rf ../lib/obj_$NAME; mkdir -p ../lib/obj_$NAME
cp -p ../lib/obj_diag/*.o ../lib/obj_$NAME/
for f in "$@"; do rm -f ../lib/obj_$NAME/$f.o; done
make -s -j8 OUT=../lib/libcomms_hip_$NAME.so OBJDIR=../lib/obj_$NAME EXTRA="-DCOMMS_DIAG $FLAGS" $MAKEVARS 2>&1 | grep -v load-store-opt || true
ls -la ../lib/libcomms_hip_$NAME.so
