#!/bin/bash
# three-handle throughput of several builds on ONE box (3 alternations): tools/ab_kbench3.sh liborbx_a.so liborbx_b.so ...
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do for L in "$@"; do ORBX_LIB_NAME=$L KB_TAG=3h_$L KB_HANDLES=3 KB_NOPROF=1 python tools/kbench.py 32 300 | grep pairs; done; done
