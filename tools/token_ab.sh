cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  KB_TAG="token" KB_HANDLES=3 KB_NOPROF=1 python tools/kbench.py 32 300
  ORBX_NO_DETECT_TOKEN=1 KB_TAG="no token" KB_HANDLES=3 KB_NOPROF=1 python tools/kbench.py 32 300
  ORBX_NO_DETECT_TOKEN=1 KB_TAG="no token h4" KB_HANDLES=4 KB_NOPROF=1 python tools/kbench.py 32 300
  KB_TAG="token h4" KB_HANDLES=4 KB_NOPROF=1 python tools/kbench.py 32 300
  KB_TAG="token h2" KB_HANDLES=2 KB_NOPROF=1 python tools/kbench.py 32 300
done
