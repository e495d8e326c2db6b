#!/bin/bash
# duration + FETCH_SIZE of the remap kernel for several builds: tools/remap_pmc_ab.sh liborbx.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  RG=${RG:-0} $R/tools/remap_ab.sh $L
  rm -rf /tmp/ppf
  ORBX_LIB_NAME=$L ORBX_REMAP_GROUP=${RG:-0} timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/ppf -o pmc -- python $R/tools/bench_preproc.py rectify > /dev/null 2>&1
  python $R/tools/pmc_insts.py $(find /tmp/ppf -name "*.db" | head -1) k_remap
done
