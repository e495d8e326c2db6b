#!/bin/bash
# k_remap_lds / k_remap1 durations at 64 x 1280x720 for several builds on ONE box: tools/remap_ab.sh liborbx.so[:hook] ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  L=${spec%%:*}; hook=1; [ "$spec" != "$L" ] && hook=${spec#*:}
  rm -rf /tmp/ppx
  ORBX_LIB_NAME=$L ORBX_REMAP_LDS=$hook ORBX_REMAP_GROUP=${RG:-0} rocprofv3 --kernel-trace --stats -d /tmp/ppx -o pp -- python $R/tools/bench_preproc.py rectify > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/ppx -name "*.db" | head -1) /tmp/pp.csv > /dev/null
  echo "$spec $(grep -E 'remap' /tmp/pp.csv | sed 's/(orbx::RemapArgs[^"]*)//')"
done
