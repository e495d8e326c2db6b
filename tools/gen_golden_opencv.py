#!/usr/bin/env python3
"""ONE COMMAND that pins every OpenCV rule the oracle restates from memory ("[OCV-mem]") against
real OpenCV.  This image has no cv2 (no network), so the fixture cannot be produced here; on any box with
`opencv-python` / `opencv-python-headless` >= 4.5.1 (numpy only besides; a 4.0 .. 4.5.0 wheel works too -- the tests then report which
of the blur models 440 / 44016 / 44032 of orbx_set_opencv_compat that build follows):

    python tools/gen_golden_opencv.py            # writes tests/golden/opencv_pins.npz
    python -m pytest tests/test_oracle_vs_opencv.py tests/test_gpu_parity.py -q

tests/test_oracle_vs_opencv.py is skipped while the fixture is absent and compares the oracle with it bit for bit
(float results: to the stated tolerance) once it exists.  Every entry names the reference call site it pins:

  resize      cv::resize(INTER_LINEAR) 8U            src/ORBextractor.cc:1122 (pyramid chain), src/System.cc:297-298
  fast        cv::FAST(img, kps, t, true) TYPE_9_16  src/ORBextractor.cc:810-826 (positions, order, responses)
  blur        cv::GaussianBlur(7x7, 2, 2, REFLECT_101) src/ORBextractor.cc:1075
  atan2       cv::fastAtan2                          src/ORBextractor.cc:98
  knn         cv::BFMatcher(NORM_HAMMING).knnMatch   src/Frame.cc:46,1293
  remap       cv::remap(INTER_LINEAR)                src/System.cc:294-295
  clahe       cv::createCLAHE(3.0, (8,8))->apply     Examples/Stereo/stereo_tum_vi.cc:100,142-143
  gray        cv::cvtColor(RGB2GRAY / BGR2GRAY)      src/Tracking.cc:1394-1412
  undistort   cv::undistortPoints(pts, K, D, R=I, P=K) src/Frame.cc:869,900

Inputs are the images / arrays already committed under tests/golden/ (data, no reference code).  IPP / OpenCL are
switched off where the build allows it (IPP's ippiResizeLinear differs by +-1 from the generic path the oracle restates);
the build information is stored with the fixture so that a mismatch can be attributed.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def pyramid_sizes(w, h, nlevels=8, sf=1.2):
    """(w_l, h_l) exactly as ORBextractor::ComputePyramid derives them (float32 tables, cvRound)."""
    scale = [np.float32(1.0)]
    for _ in range(1, nlevels):
        scale.append(np.float32(np.float64(scale[-1]) * np.float64(np.float32(sf))))
    out = []
    for s in scale:
        inv = np.float32(1.0) / s
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
    return out


def main():
    try:
        import cv2
    except ImportError:
        print("cv2 is not installed here: run this script on a box with opencv-python >= 4.5.1", file=sys.stderr)
        return 2
    try:
        cv2.ipp.setUseIPP(False)
    except Exception:
        pass
    try:
        cv2.ocl.setUseOpenCL(False)
    except Exception:
        pass
    cv2.setNumThreads(1)
    rng = np.random.default_rng(20220131)
    pin = np.load(os.path.join(G, "fast9_skimage.npz"))
    rc = np.load(os.path.join(G, "rectify_clahe.npz"))
    ud = np.load(os.path.join(G, "undistort.npz"))
    st = np.load(os.path.join(G, "stereo_400x300.npz"))
    out = {"cv_version": np.array(cv2.__version__), "cv_build": np.array(cv2.getBuildInformation()[:4000]),
           "use_ipp": np.array(bool(getattr(cv2.ipp, "useIPP", lambda: False)()))}
    imgs = {n: pin["img_" + n] for n in ("g384", "g400L", "g160")}

    # ---- resize: the pyramid chain (level l from level l-1) + one arbitrary ratio + exact 2x (INTER_AREA shortcut)
    for n, im in imgs.items():
        sizes = pyramid_sizes(im.shape[1], im.shape[0], 3 if n == "g160" else 8)
        cur = im
        for l in range(1, len(sizes)):
            cur = cv2.resize(cur, sizes[l], interpolation=cv2.INTER_LINEAR)
            out["resize_%s_L%d" % (n, l)] = cur
    out["resize_g384_to_251x97"] = cv2.resize(imgs["g384"], (251, 97), interpolation=cv2.INTER_LINEAR)
    out["resize_g384_half"] = cv2.resize(imgs["g384"], (192, 144), interpolation=cv2.INTER_LINEAR)
    out["resize_g384_up"] = cv2.resize(imgs["g384"], (500, 333), interpolation=cv2.INTER_LINEAR)

    # ---- FAST 9-16 with NMS: whole images and cell-sized ROIs at both thresholds (position order + response)
    for n, im in imgs.items():
        for t in (20, 7):
            det = cv2.FastFeatureDetector_create(threshold=t, nonmaxSuppression=True,
                                                 type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
            kps = det.detect(im, None)
            out["fast_%s_t%d" % (n, t)] = np.array([[k.pt[0], k.pt[1], k.response] for k in kps], np.float32).reshape(-1, 3)
            det0 = cv2.FastFeatureDetector_create(threshold=t, nonmaxSuppression=False,
                                                  type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
            out["fastraw_%s_t%d" % (n, t)] = np.array([[k.pt[0], k.pt[1]] for k in det0.detect(im, None)], np.float32).reshape(-1, 2)
    rois = []
    big = imgs["g400L"]
    for _ in range(12):
        hh, ww = int(rng.integers(41, 73)), int(rng.integers(41, 54))
        y, x = int(rng.integers(0, big.shape[0] - hh)), int(rng.integers(0, big.shape[1] - ww))
        rois.append((y, x, hh, ww))
        roi = big[y:y + hh, x:x + ww]   # a VIEW with the parent's stride, as the reference passes cell ROIs
        det = cv2.FastFeatureDetector_create(threshold=20, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
        kps = det.detect(roi, None)
        out["fastroi_%d" % (len(rois) - 1)] = np.array([[k.pt[0], k.pt[1], k.response] for k in kps], np.float32).reshape(-1, 3)
    out["fastroi_rects"] = np.array(rois, np.int32)

    # ---- GaussianBlur 7x7 sigma 2 (whole level, not a sub-matrix: the reference blurs a clone)
    for n, im in imgs.items():
        out["blur_" + n] = cv2.GaussianBlur(im.copy(), (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
    imp = np.zeros((15, 15), np.uint8)
    imp[7, 7] = 255
    out["blur_impulse"] = cv2.GaussianBlur(imp, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
    out["gauss_kernel_7_2"] = cv2.getGaussianKernel(7, 2).ravel()

    # ---- fastAtan2 (scalar entry point == cv::fastAtan2(float y, float x))
    yx = np.concatenate([rng.integers(-2000000, 2000000, (4000, 2)), rng.integers(-300, 300, (2000, 2)),
                         np.array([[0, 0], [0, 1], [1, 0], [0, -1], [-1, 0], [1, 1], [-1, -1], [1, -1], [-1, 1]])]).astype(np.float32)
    out["atan2_yx"] = yx
    out["atan2_deg"] = np.array([cv2.fastAtan2(float(y), float(x)) for y, x in yx], np.float32)

    # ---- BFMatcher(NORM_HAMMING).knnMatch k = 2 on committed descriptors (ties by train index)
    q, t = st["dL"], st["dR"]
    m = cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(q, t, k=2)
    out["knn_idx"] = np.array([[mm[0].trainIdx, mm[1].trainIdx] for mm in m], np.int32)
    out["knn_dist"] = np.array([[mm[0].distance, mm[1].distance] for mm in m], np.float32)
    qd = np.repeat(q[:40], 3, axis=0)   # heavy ties: repeated rows
    m = cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(q[:60], qd, k=2)
    out["knn_tie_train"] = qd
    out["knn_tie_idx"] = np.array([[mm[0].trainIdx, mm[1].trainIdx] for mm in m], np.int32)

    # ---- remap / CLAHE / cvtColor on the committed pre-processing inputs
    out["remap"] = cv2.remap(rc["img"], rc["map_x"], rc["map_y"], cv2.INTER_LINEAR)
    out["clahe_3_8x8"] = cv2.createCLAHE(3.0, (8, 8)).apply(rc["img"])
    out["clahe_2_4x3"] = cv2.createCLAHE(2.0, (4, 3)).apply(rc["img"])
    out["clahe_g384"] = cv2.createCLAHE(3.0, (8, 8)).apply(imgs["g384"])
    col = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    out["gray_in"] = col
    out["gray_rgb"] = cv2.cvtColor(col, cv2.COLOR_RGB2GRAY)
    out["gray_bgr"] = cv2.cvtColor(col, cv2.COLOR_BGR2GRAY)
    col4 = rng.integers(0, 256, (32, 48, 4), dtype=np.uint8)
    out["gray4_in"] = col4
    out["gray_rgba"] = cv2.cvtColor(col4, cv2.COLOR_RGBA2GRAY)
    out["gray_bgra"] = cv2.cvtColor(col4, cv2.COLOR_BGRA2GRAY)

    # ---- undistortPoints with P = K (Frame::UndistortKeyPoints) on the committed keypoints
    for rig in ("euroc", "tum1"):
        K4, D = ud[rig + "_K"], ud[rig + "_D"]
        K = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1]], np.float32)
        kp = ud[rig + "_kps"].copy().view(np.float32).reshape(-1, 7)[:, :2].astype(np.float32)
        un = cv2.undistortPoints(kp.reshape(-1, 1, 2), K, D.astype(np.float32), R=np.eye(3, dtype=np.float32), P=K)
        out["undistort_" + rig] = un.reshape(-1, 2).astype(np.float32)

    np.savez_compressed(os.path.join(G, "opencv_pins.npz"), **out)
    print("OpenCV", cv2.__version__, "-> tests/golden/opencv_pins.npz,", len(out), "entries")
    return 0


if __name__ == "__main__":
    sys.exit(main())
