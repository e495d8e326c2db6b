#!/bin/bash
# A/B of the even-ring survivor filter (liborbx_even.so = the same sources with -DORBX_EVEN_FILTER=1), on ONE box
cd ${GRAFT_REPO_ROOT:-.}
ORBX_LIB_NAME=liborbx_even.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pin_skimage.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
for i in 1 2 3; do
  ORBX_LIB_NAME=liborbx_base.so KB_TAG=base python tools/kbench.py 32 30 | grep -E "k_detect|step|pairs"
  ORBX_LIB_NAME=liborbx_even.so KB_TAG=even python tools/kbench.py 32 30 | grep -E "k_detect|step|pairs"
done
for i in 1 2 3; do
  ORBX_LIB_NAME=liborbx_base.so KB_TAG=base3 KB_HANDLES=3 KB_NOPROF=1 python tools/kbench.py 32 300 | tail -1
  ORBX_LIB_NAME=liborbx_even.so KB_TAG=even3 KB_HANDLES=3 KB_NOPROF=1 python tools/kbench.py 32 300 | tail -1
done
python tools/cell_stats.py 8
