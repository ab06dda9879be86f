"""GPU: the Modeler seam end to end.  adapters/semi_dense_queue.h instantiated with the REAL
ProbabilityMapping -- SemiDenseQueue -- is driven like the fork's Modeler thread would drive it
(/root/reference/src/Modeler/Modeler.cc:100-128, 1465-1472: enqueue per new keyframe, one keyframe per idle pass, pinned
while worked on) over test doubles of the fork's KeyFrame / MapPoint / cv::Mat, and every point the Injector receives
(what addKeyFrameInsertionWithLinesEntry would write, SFMTranscriptInterface_ORBSLAM.cpp:319-374) is compared bit for
bit with the same schedule replayed on the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

from common import Sequence, assert_bit_equal
from test_gpu_cpp_class import write_blob

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_queue_driver(pkg):
    pkg.build_mod.build_all()
    lib = os.path.join(ROOT, "orb-slam-free-space-carving_amd", "lib")
    exe = os.path.join(lib, "test_semi_dense_queue_gpu")
    src = os.path.join(ROOT, "tests", "cpp", "test_semi_dense_queue_gpu.cc")
    deps = [src, os.path.join(lib, "libsdm_pm.so"), os.path.join(ROOT, "adapters", "semi_dense_queue.h"),
            os.path.join(ROOT, "adapters", "orbslam_carv_adapter.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror",
                               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "adapters"),
                               "-I" + os.path.join(ROOT, "tests", "cpp", "mock_fork"), src, "-o", exe,
                               "-L" + lib, "-lsdm_pm", "-lsdm_hip", "-Wl,-rpath," + lib])
    return exe


def test_queue_with_real_mapper_injects_the_oracle_points(pkg, oracle, gpu_ok, tmp_path):
    exe = build_queue_driver(pkg)
    n_kf, n = 16, 7
    seq = Sequence(pkg, oracle, 96, 72, n_kf, 0x5EED0F03)
    W, H = seq.W, seq.H
    rng = np.random.default_rng(3)
    depths = [(1.0 + 0.1 * rng.standard_normal(120)).astype(np.float32) for _ in range(n_kf)]
    blob, out = tmp_path / "in.bin", tmp_path / "out.bin"
    write_blob(blob, seq, n_kf, n, depths)
    # one-sided covisibility (a new keyframe only sees older ones) leaves few points under the obj writer's sigma <= 0.01;
    # the filter is the queue's parameter, 0.1 here so that the comparison sees a few thousand points
    max_sigma = 0.1
    subprocess.check_call([exe, str(blob), str(out), repr(max_sigma)])
    raw = np.fromfile(out, dtype=np.uint8)
    off = 0

    def take(dtype, count):
        nonlocal off
        a = raw[off:off + count * np.dtype(dtype).itemsize].view(dtype)
        off += count * np.dtype(dtype).itemsize
        return a

    injected = []
    while True:
        k, npts = take(np.int32, 2)
        if k < 0:
            break
        injected.append((int(k), take(np.float32, 3 * int(npts)).reshape(-1, 3)))
    got_depths, flags = [], []
    for k in range(n_kf):
        nd = int(take(np.int32, 1)[0])
        got_depths.append(take(np.float32, nd).copy())
        flags.append(tuple(int(v) for v in take(np.int32, 2)))
    processed, lookups, pin_errors, unpinned, n_map = (int(v) for v in take(np.int32, 5))
    assert processed == n_kf and lookups == n_kf and n_map == n_kf
    assert pin_errors == 0 and unpinned == n_kf  # every injection inside SetNotErase/SetErase, all pins released
    for k in range(n_kf):  # the adapter's camera depths of the map points (the loop of src/KeyFrame.cc:644-662)
        assert len(got_depths[k]) == len(depths[k]) and np.allclose(got_depths[k], depths[k], rtol=0, atol=1e-5)

    # ---- the same schedule on the oracle: keyframes arrive in creation order; a keyframe sees the covisible keyframes
    # that are registered by then (the first n of them, PM.cc:151-160); after each SemiDenseRecon every keyframe whose
    # neighbours are all reconstructed is checked IN PLACE, in map order (PM.cc:262-315), and handed over once
    full_cov = {k: seq.scene.neighbours(k, n_kf, n_kf - 1) for k in range(n_kf)}
    rho, sig, xyz = {}, {}, {}
    semi, inter = [False] * n_kf, [False] * n_kf
    want = []
    for k in range(n_kf):
        reg = set(range(k + 1))
        nb = [j for j in full_cov[k] if j in reg][:n]
        if len(nb) < n:
            continue  # PM.cc:160: not enough neighbours yet; the call returns before the inter-keyframe phase, too
        mn, mx = oracle.stereo_search_constraints(got_depths[k])
        rho[k], sig[k], _ = oracle.semi_dense_recon(seq.okf[k], [seq.okf[j] for j in nb], None, mn, mx)
        semi[k] = True
        newly = []
        for i in range(k + 1):
            if inter[i] or not semi[i]:
                continue
            nbi = [j for j in full_cov[i] if j in reg][:n]
            if len(nbi) < n or not all(semi[j] for j in nbi):
                continue
            rho[i] = oracle.inter_check(seq.okf[i], rho[i], [seq.okf[j] for j in nbi], [rho[j] for j in nbi],
                                        [sig[j] for j in nbi])
            xyz[i] = oracle.pointset(seq.okf[i], rho[i])
            inter[i] = True
            newly.append(i)
        for i in sorted(newly):
            m = (sig[i] <= max_sigma) & (rho[i] > 1e-6)
            want.append((i, xyz[i].reshape(H, W, 3)[m]))
    assert [k for k, _ in injected] == [k for k, _ in want]
    assert len(want) >= 3 and sum(len(p) for _, p in want) > 50, "the schedule must reach the mesher with points"
    for (k, got), (_, pts) in zip(injected, want):
        assert_bit_equal(got, np.ascontiguousarray(pts, dtype=np.float32), "injected points of keyframe %d" % k)
    assert flags == [(int(semi[k]), int(inter[k])) for k in range(n_kf)]
