#!/bin/bash
# Probe a box for any usable OpenCV (python module, shared libraries, headers, pip index).  Output is committed under
# profiles/ as the evidence for DESIGN.md 2 ("parity unpinned": no OpenCV on either box).
out=${1:-gpurun_out/r3_opencv_probe.txt}
{
  echo "== date: $(date -u)"; echo "== host: $(uname -a)"
  for py in python python3 /usr/bin/python3 /opt/conda/bin/python /opt/conda/bin/python3.9; do
    echo "== $py -c 'import cv2'"; $py -c "import cv2; print(cv2.__version__); print(cv2.getBuildInformation())" 2>&1 | head -60
  done
  echo "== find cv2 / libopencv / opencv headers"
  find / \( -name "cv2*" -o -name "libopencv*" -o -name "opencv2" -o -name "opencv4" \) -not -path "/proc/*" 2>/dev/null | head -20
  echo "== pkg-config"; pkg-config --modversion opencv4 2>&1 | head -3
  echo "== pip download opencv-python-headless (index reachability)"
  timeout 60 python -m pip download --no-deps -d /tmp/ocv_wheel opencv-python-headless 2>&1 | tail -5
  echo "== conda"; timeout 60 /opt/conda/bin/conda install -y --dry-run opencv 2>&1 | tail -5
  echo "== network"; timeout 10 python - <<'PY' 2>&1 | tail -3
import socket
for host in ("pypi.org", "files.pythonhosted.org", "github.com"):
    try:
        print(host, socket.getaddrinfo(host, 443)[0][4])
    except Exception as e:
        print(host, "unreachable:", e)
PY
} > "$out" 2>&1
echo "probe written to $out"
