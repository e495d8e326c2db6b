#!/bin/bash
# quick A/B on ONE box: liborbx_base.so (the build to compare against) vs liborbx.so -- kbench one handle (stage times) and three handles
cd ${GRAFT_REPO_ROOT:-.}
if [ -n "${TESTS:-}" ]; then timeout 900 python -m pytest $TESTS -m gpu -x -q 2>&1 | grep -E "passed|failed"; fi
for i in 1 2 3; do
  ORBX_LIB_NAME=liborbx_base.so KB_TAG=base python tools/kbench.py 32 30 | grep -E "pairs"
  KB_TAG=new python tools/kbench.py 32 30 | grep -E "pairs"
done
for i in 1 2 3; do
  ORBX_LIB_NAME=liborbx_base.so KB_TAG=base3 KB_HANDLES=3 KB_NOPROF=1 python tools/kbench.py 32 300 | tail -1
  KB_TAG=new3 KB_HANDLES=3 KB_NOPROF=1 python tools/kbench.py 32 300 | tail -1
done
