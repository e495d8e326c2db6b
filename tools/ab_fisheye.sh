#!/bin/bash
# one-handle C4 kernel durations of several builds on ONE box: tools/ab_fisheye.sh liborbx_a.so liborbx_b.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  rm -rf /tmp/cp_kt
  ORBX_LIB_NAME=$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp_kt -o kt -- python $R/bench.py --no-extras --no-profile --handles 1 --steps 20 --warmup 3 --mode fisheye --width 512 --height 512 --nfeatures 1500 > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/cp_kt -name "*.db" | head -1) /tmp/ks.csv > /dev/null
  echo "$L $(grep -E 'k_fisheye_(batch|scan|tri)' /tmp/ks.csv | cut -d, -f1,4 | tr '\n' ' ')"
done
