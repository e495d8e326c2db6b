#!/bin/bash
# A/B of the single-frame C call on ONE box: liborbx_base.so (build of the commit to compare against) vs liborbx.so, alternating
#   tools/lat_ab.sh [rounds=4] [calls=500]
cd ${GRAFT_REPO_ROOT:-.}
for r in $(seq 1 ${1:-4}); do
  ORBX_PROF_LIB=liborbx_base.so LAT_TAG=base python tools/lat_c.py ${2:-500}
  LAT_TAG=new python tools/lat_c.py ${2:-500}
done | tee /tmp/lat_ab.txt
python - <<'PY'
import re
b, n = [], []
for l in open("/tmp/lat_ab.txt"):
    m = re.search(r"mean ([0-9.]+) p50 ([0-9.]+)", l)
    if m: (b if l.strip().endswith("base") else n).append((float(m.group(1)), float(m.group(2))))
f = lambda v, k: sum(x[k] for x in v) / len(v)
print("base mean %.4f p50 %.4f | new mean %.4f p50 %.4f | delta mean %+.1f us, p50 %+.1f us" % (f(b, 0), f(b, 1), f(n, 0), f(n, 1), 1e3 * (f(n, 0) - f(b, 0)), 1e3 * (f(n, 1) - f(b, 1))))
PY
