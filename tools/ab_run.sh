#!/bin/bash
# A/B of two builds on ONE box: tools/ab_run.sh  (liborbx_base.so = the build to compare against, see ORBX_LIB_NAME)
set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
if [ -z "${NOTEST:-}" ]; then timeout 900 python -m pytest ${TESTS:-tests/test_gpu_parity.py} -m gpu -x -q > gpurun_out/ab_parity.log 2>&1; echo "parity rc=$?"; tail -2 gpurun_out/ab_parity.log; fi
for i in 1 2 3; do
  ORBX_LIB_NAME=liborbx_base.so KB_TAG=base python tools/kbench.py 32 30
  KB_TAG=new python tools/kbench.py 32 30
done
for i in 1 2 3 4; do
  ORBX_LIB_NAME=liborbx_base.so KB_TAG=base3 KB_HANDLES=3 KB_NOPROF=1 python tools/kbench.py 32 300
  KB_TAG=new3 KB_HANDLES=3 KB_NOPROF=1 python tools/kbench.py 32 300
done
