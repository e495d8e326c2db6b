#!/bin/bash
# 640x480 mono, 16 frames per step (config C2 at the small batch of bench.py's other_configs): level kernels against cascade plans
# (ORBX_LAT_MAX_IMAGES=16, ORBX_LAT_TAIL=levels,rows of the last level per workgroup)
cd ${GRAFT_REPO_ROOT:-.}
summ='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-14s %10.1f %s  %7.4f ms/step" % (sys.argv[1], d["value"], d["unit"], d["ms_per_step"]))'
run() { tag=$1; shift; timeout 300 python bench.py --no-extras --cpu-pairs 0 --mode mono --width 640 --height 480 --nfeatures 1000 --pairs 8 --steps 400 --no-profile 2>/dev/null | python -c "$summ" "$tag"; }
for rep in 1 2; do
  run "level kernels"
  for lt in 2,4 2,8 2,16 3,4 3,8 4,2 4,4 4,8; do
    ORBX_LAT_MAX_IMAGES=16 ORBX_LAT_TAIL=$lt run "cascade $lt"
  done
done
