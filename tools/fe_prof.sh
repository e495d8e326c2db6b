R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cp_kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/cp_kt -o kt -- python $R/bench.py --no-extras --handles 1 --mode fisheye --width 512 --height 512 --nfeatures 1500 > /tmp/b.json 2>/dev/null
python $R/tools/rocprof_summary.py $(find /tmp/cp_kt -name "*.db" | head -1) /tmp/ks.csv > /dev/null; head -12 /tmp/ks.csv
