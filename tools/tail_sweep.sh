#!/bin/bash
# k_resize_tail plan sweep on one box: ORBX_RESIZE_TAIL=first,levels,rows (tools/tail_sweep.sh)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do
for cfg in "" "5,3,13" "5,3,18" "5,3,9" "6,2,26" "5,2,26"; do
  ORBX_RESIZE_TAIL=$cfg KB_TAG="3h tail[$cfg]" KB_HANDLES=3 KB_NOPROF=1 python tools/kbench.py 32 300
done
done
