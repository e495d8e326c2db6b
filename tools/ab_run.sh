set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/ab1_parity.log 2>&1; echo "parity rc=$?" 
for i in 1 2 3; do
  ORBX_LIB_NAME=liborbx_base.so KB_TAG=base python tools/kbench.py 32 20
  KB_TAG=new python tools/kbench.py 32 20
done
for i in 1 2; do
  ORBX_LIB_NAME=liborbx_base.so KB_TAG=base3 KB_HANDLES=3 KB_NOPROF=1 python tools/kbench.py 32 60
  KB_TAG=new3 KB_HANDLES=3 KB_NOPROF=1 python tools/kbench.py 32 60
done
tail -3 gpurun_out/ab1_parity.log
