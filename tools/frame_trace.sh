#!/bin/bash
# Timeline of one single-frame orbx_extract_stereo call (run on the GPU box):  tools/frame_trace.sh [out.txt] [script.py]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/frame_trace.txt}; case "$OUT" in /*) ;; *) OUT=$R/$OUT;; esac; mkdir -p "$(dirname "$OUT")"
SCRIPT=${2:-$R/tools/lat_trace.py}
D=$(mktemp -d /tmp/ftrace.XXXXXX)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d $D -o ft -- python $SCRIPT > $D/run.log 2>&1
cd $R
DB=$(find $D -name "*.db" | head -1)
python tools/frame_trace.py $DB 3 > $OUT 2>&1
tail -5 $D/run.log >> $OUT
rm -rf $D
