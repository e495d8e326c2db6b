#!/bin/bash
# one-handle stage times of several builds on ONE box (3 alternations): tools/ab_kbench1.sh liborbx_a.so liborbx_b.so ...
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do for L in "$@"; do ORBX_LIB_NAME=$L KB_TAG=$L python tools/kbench.py 32 30 | grep pairs; done; done
