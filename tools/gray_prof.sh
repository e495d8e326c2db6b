R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 300 python -m pytest tests/test_preprocess.py tests/test_rectify_clahe.py -m gpu -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ppg
rocprofv3 --kernel-trace --stats -d /tmp/ppg -o pp -- python $R/tools/bench_preproc.py gray > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/ppg -name "*.db" | head -1) $R/gpurun_out/r6_preproc_gray_stats.csv > /dev/null; cat $R/gpurun_out/r6_preproc_gray_stats.csv | cut -c1-140
for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/ppg2; rocprofv3 --pmc $c --kernel-trace -d /tmp/ppg2 -o pmc -- python $R/tools/bench_preproc.py gray > /dev/null 2>&1; python $R/tools/pmc_insts.py $(find /tmp/ppg2 -name "*.db" | head -1) k_cvt; done
