#!/bin/bash
# Developer aid: build libskdsp_hip_<name>.so with one translation unit recompiled with extra flags
#   tools/build_variant.sh <name> <file.hip> "<flags>"      (run from the repo root; use via SKDSP_LIB=<path>)
set -e
NAME=$1; SRC=$2; FLAGS=$3
cd "$(dirname "$0")/../scikit-dsp-comm_amd/csrc"
EXTRA=""
[ "$SRC" = fir_ols.hip ] && EXTRA="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../include $EXTRA $FLAGS -c $SRC -o build/${SRC%.hip}_$NAME.o
OBJS=""
# (iir_scan.hip is two objects in the product build -- SK_SCAN_PART; a variant of it is one object holding both parts)
LIST="capi dist fir_bx fir_direct fir_mm fir_ols fir_ols64 iir_fused iir_par iir_scan resample"
[ "$SRC" = iir_scan.hip ] || LIST="$LIST iir_scan_f64"
for f in $LIST; do
  if [ "$f.hip" = "$SRC" ]; then OBJS="$OBJS build/${f}_$NAME.o"; else OBJS="$OBJS build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../sk_dsp_comm_amd/libskdsp_hip_$NAME.so $OBJS -ldl
echo built ../sk_dsp_comm_amd/libskdsp_hip_$NAME.so
