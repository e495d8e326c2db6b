#!/bin/bash
# Build A/B variants of liborbx that differ in ONE translation unit's flags:
#   tools/build_variant.sh <unit.hip> name1:"-DFLAG=1" name2:"-DX -DY" ...   -> orb_slam3_fast_amd/liborbx_<name>.so
# (the other objects are the ones `make` left in csrc/; run make first)
set -u
cd "$(dirname "$0")/../orb_slam3_fast_amd/csrc" || exit 1
UNIT=$1; shift
OBJ=${UNIT%.hip}.o
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Wno-unused-const-variable -Wno-unused-variable"
OTHERS=$(ls *.o | grep -v "^$OBJ$")
mkdir -p /tmp/orbx_variants
for v in "$@"; do
  n=${v%%:*}; fl=${v#*:}
  ( /opt/rocm/bin/hipcc $F $fl -c -o /tmp/orbx_variants/${n}_$OBJ $UNIT 2>&1 | grep -E "error|Spill: [1-9]" ;
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../liborbx_$n.so /tmp/orbx_variants/${n}_$OBJ $OTHERS -ldl -Wl,-rpath,/opt/rocm/lib ) &
done
wait
ls ../liborbx_*.so
