"""CPU: adapters/orbslam_carv_adapter.h compiles against test doubles of the fork's
KeyFrame / MapPoint and of the cv::Mat operations it uses (tests/cpp/mock_fork -- NOT the real headers, which
the image does not have) and fills sdm::KeyFrame as INTEGRATION.md §2 says.  Host logic only; no GPU, no oracle."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adapter_fills_semi_dense_keyframe(tmp_path):
    exe = os.path.join(str(tmp_path), "test_adapter")
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-O1",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "adapters"),
           "-I" + os.path.join(ROOT, "tests", "cpp", "mock_fork"),
           os.path.join(ROOT, "tests", "cpp", "test_adapter.cc"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
