"""Builds tests/cpp/frame_like.cpp -- the C++ program that drives the drop-in classes (csrc/ORBextractor.h, ORBmatcher.h, ...) the way
the reference's Frame does -- against the shipped liborbx.so.  Used by tests/test_cpp_mirror.py and by bench.py's latency leg."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "frame_like")
STUB = os.path.join(ROOT, "tests", "cpp", "opencv_stub")


def build_exe(cv=False):
    """cv=False: the cvlite stand-in types (-DORBX_NO_OPENCV).  cv=True: the `#ifdef ORBX_HAVE_OPENCV` branch of the mirror
    headers -- the reference's own signatures, int operator()(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint>&,
    cv::OutputArray, std::vector<int>&) (include/ORBextractor.h:64-68) -- compiled against tests/cpp/opencv_stub (a test-only
    model of the few <opencv2/core.hpp> members those branches touch; the image has no OpenCV)."""
    src = os.path.join(ROOT, "tests", "cpp", "frame_like.cpp")
    libdir = os.path.join(ROOT, "orb_slam3_fast_amd")
    exe = EXE + ("_cv" if cv else "")
    hdrs = [os.path.join(libdir, "csrc", h) for h in ("ORBextractor.h", "ORBmatcher.h", "Preprocess.h", "ORBVocabulary.h")]
    hdrs.append(os.path.join(STUB, "opencv2", "core.hpp"))
    if (not os.path.exists(exe)) or any(os.path.getmtime(p) > os.path.getmtime(exe) for p in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I" + STUB if cv else "-DORBX_NO_OPENCV", src, "-o", exe,
                               "-L" + libdir, "-lorbx", "-lpthread", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe
