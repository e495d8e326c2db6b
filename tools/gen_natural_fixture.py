#!/opt/conda/bin/python3.9
"""Natural-image inputs for the parity tests (round-2 VERDICT item 6): every other test frame comes from synth.py.

The reference's configs run on EuRoC / TUM-VI / ZED2 imagery (Examples/Monocular/EuRoC.yaml:33-63,
Examples/Stereo-Inertial/TUM-VI.yaml:45-49,86) that is not on disk; the build container's scikit-image 0.18.3 ships natural
photographs AND a rectified stereo pair:
  camera            512 x 512 grey   (= BASELINE config C4's frame size)
  astronaut         512 x 512 RGB    -> grey
  motorcycle_left / motorcycle_right   741 x 500 RGB -> grey: a rectified Middlebury-2014 stereo pair (disparity along x)
RGB -> grey is cv::cvtColor(RGB2GRAY)'s fixed point, (R*4899 + G*9617 + B*1868 + 8192) >> 14, the conversion of
Tracking::GrabImage* (src/Tracking.cc:1394-1412).
Run in THIS container only (the GPU box has no scikit-image):   /opt/conda/bin/python3.9 tools/gen_natural_fixture.py
Writes tests/golden/natural_images.npz (pixel arrays only).
"""
import hashlib
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
import skimage  # noqa: E402
from skimage import io  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(os.path.dirname(skimage.__file__), "data")


def grey(a):
    if a.ndim == 2:
        return np.ascontiguousarray(a, np.uint8)
    a = a[..., :3].astype(np.uint32)
    return ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def main():
    out = {}
    for key, f in (("camera", "camera.png"), ("astronaut", "astronaut.png"), ("moto_left", "motorcycle_left.png"),
                   ("moto_right", "motorcycle_right.png")):
        out[key] = grey(io.imread(os.path.join(D, f)))
        print(key, out[key].shape, hashlib.sha256(out[key].tobytes()).hexdigest()[:16])
    out["source"] = np.array("scikit-image %s skimage/data/{camera,astronaut,motorcycle_left,motorcycle_right}.png, "
                             "grey = (R*4899 + G*9617 + B*1868 + 8192) >> 14" % skimage.__version__)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "natural_images.npz"), **out)


if __name__ == "__main__":
    main()
