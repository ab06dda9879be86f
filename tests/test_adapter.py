"""CPU: adapters/orbslam_carv_adapter.h compiles against test doubles of the fork's
KeyFrame / MapPoint and of the cv::Mat operations it uses (tests/cpp/mock_fork -- NOT the real headers, which
the image does not have) and fills sdm::KeyFrame as INTEGRATION.md §2 says.  Host logic only; no GPU, no oracle."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(tmp_path, name):
    exe = os.path.join(str(tmp_path), name)
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-O1",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "adapters"),
           "-I" + os.path.join(ROOT, "tests", "cpp", "mock_fork"),
           os.path.join(ROOT, "tests", "cpp", name + ".cc"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_adapter_fills_semi_dense_keyframe(tmp_path):
    _build_and_run(tmp_path, "test_adapter")


def test_semi_dense_queue_seam(tmp_path):
    """adapters/semi_dense_queue.h: the Modeler seam (queue bound/order, bad keyframes, SetNotErase/SetErase around the
    work, frame lookup, map view + mutual covisibility, one transcript injection per finished keyframe with the
    sigma/rho filter, erase and bundle-adjustment hooks), following src/Modeler/Modeler.cc:100-128, 1321-1353,
    1465-1472 -- compiled against the fork doubles with a recording mapper"""
    _build_and_run(tmp_path, "test_semi_dense_queue")
