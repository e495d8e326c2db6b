#!/bin/bash
# Secondary numbers for the BASELINE.json configs next to the headline metric (run through gpurun from the repo root):
#   tools/bench_configs.sh > gpurun_out/configs.txt
# C1 = CPU oracle on 752x480 mono (no GPU), C2 = 640x480 mono, C3 = 1280x720 stereo (headline, bench.py default),
# C4 = 512x512 fisheye stereo (lapping areas + 2-NN + KB8 triangulation), plus 640x480 stereo and 752x480 mono.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
summ='import json,sys; d=json.loads(sys.stdin.read()); print("%-8s %10.1f %-16s %7.4f ms/step  kp/img %6.1f  dominant %s %.1f us" % (sys.argv[1], d["value"], d["unit"], d["ms_per_step"], d["config"]["keypoints_per_image"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"]))'
run() { tag=$1; shift; timeout 300 python bench.py --no-extras "$@" 2>/dev/null | python -c "$summ" $tag; }
run C2      --mode mono --width 640 --height 480 --nfeatures 1000
run C1size  --mode mono --width 752 --height 480 --nfeatures 1000
run 640st   --mode stereo --width 640 --height 480 --nfeatures 1000
run C4      --mode fisheye --width 512 --height 512 --nfeatures 1500
run C3      --mode stereo
run C3x1    --mode stereo --handles 1
# C1: the CPU oracle (port of the reference's serial semantics) on 752x480 mono frames, all usable cores / one core
python - <<'PY'
import os, subprocess, sys, json, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
from orb_slam3_fast_amd import synth
import bench
frames = np.stack([synth.mono_frame(752, 480, 500 + i) for i in range(4)])
pairs = np.stack([frames, frames], 1)      # cpu_bench times pairs: (L, R) = two mono frames
tmp = os.path.join(tempfile.gettempdir(), "orbx_c1.npy")
np.save(tmp, pairs)
cores = bench.usable_cores()
for procs in (cores, 1):
    r = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", tmp, "1000", "1.0", "1.0", str(procs), "12", "--extract-only"],
                       capture_output=True, text=True)
    try:
        d = json.loads(r.stdout)
        print("C1       %10.1f frames/s  CPU oracle, 752x480 mono, 1000 features, %d worker process(es), %d frames in %.1f s"
              % (d["frames_per_s"], procs, 2 * d["pairs"], d["wall_s"]))
    except Exception:
        print("C1 failed:", r.stderr.strip()[-300:])
PY
