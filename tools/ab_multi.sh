#!/bin/bash
# timing of several builds on ONE box: tools/ab_multi.sh liborbx_a.so liborbx_b.so ...   (names inside orb_slam3_fast_amd/)
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do for L in "$@"; do ORBX_LIB_NAME=$L KB_TAG=$L python tools/kbench.py 32 30; done; done
for i in 1 2 3; do for L in "$@"; do ORBX_LIB_NAME=$L KB_TAG=3h_$L KB_HANDLES=3 KB_NOPROF=1 python tools/kbench.py 32 300; done; done
