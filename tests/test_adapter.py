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


def _plan_block(pkg, tmp_path, n_all, world, n, extra=()):
    lib = os.path.dirname(pkg.lib_path())
    exe = os.path.join(str(tmp_path), "test_plan_block")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", "test_plan_block.cc"), "-o", exe,
                               "-L" + lib, "-lsdm_pm", "-lsdm_hip", "-Wl,-rpath," + lib])
    r = subprocess.run([exe, str(n_all), str(world), str(n)] + [str(e) for e in extra], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    out = []
    for line in r.stdout.strip().split("\n"):
        d = {}
        for part in line.split(";"):
            tok = part.split()
            if tok and tok[0] == "rank":
                tok = tok[2:]
            if tok:
                d[tok[0]] = [int(v) for v in tok[1:]]
        out.append(d)
    return out


def test_cpp_block_plan_equals_python_plan(pkg, tmp_path):
    """ProbabilityMapping::PlanBlock (what SemiDenseReconBlock sends, receives and reconstructs first on each rank) equals
    shard.plan -- the plan the bench and the multi-rank tests run -- on index-local covisibility, for several shapes.
    The C++ sharded driver cannot run with world > 1 on a one-GPU box; this pins its list logic on CPU."""
    nb = pkg.synth.Scene.neighbours
    for (n_all, world, n) in [(24, 2, 7), (32, 4, 6), (64, 8, 20), (36, 3, 4)]:
        got = _plan_block(pkg, tmp_path, n_all, world, n)
        assert len(got) == world
        for r, g in enumerate(got):
            pl = pkg.shard.plan(n_all, world, r, n, nb)
            assert g["refs"] == pl["own"] and g["check"] == pl["own"]
            assert g["check_nbrs"] == g["nbrs"]
            assert g["nbrs"] == [j for row in pl["nbrs"] for j in row]
            assert g["needed"] == pl["inputs"]
            assert g["boundary"] == pl["boundary"]
            send = [(p, k) for p in sorted(pl["send"]) for k in pl["send"][p]]
            recv = [(p, k) for p in sorted(pl["recv"]) for k in pl["recv"][p]]
            assert list(zip(g["send_peer"], g["send_kf"])) == send
            assert list(zip(g["recv_peer"], g["recv_kf"])) == recv
    # the k-th send of rank a to rank b is rank b's k-th receive from a, also when keyframes drop out (bad / done)
    got = _plan_block(pkg, tmp_path, 32, 4, 6, extra=(9, 17))
    for a in range(4):
        for b in range(4):
            if a != b:
                s = [k for p, k in zip(got[a]["send_peer"], got[a]["send_kf"]) if p == b]
                r_ = [k for p, k in zip(got[b]["recv_peer"], got[b]["recv_kf"]) if p == a]
                assert s == r_, (a, b, s, r_)
    assert 9 not in got[1]["refs"] and 17 not in got[2]["refs"] and 9 not in got[1]["nbrs"]


def test_cpp_block_plan_second_pass_is_pairwise_consistent(pkg, tmp_path):
    """A second sharded pass (keyframes that became usable after the first) is planned from replicated flags only: on
    every rank pair the k-th send equals the k-th receive, nobody sends or reads a map that does not exist, already
    reconstructed or checked keyframes are not processed again (PM.cc:141, 265)."""
    n_all, world, n = 40, 4, 6
    got = _plan_block(pkg, tmp_path, n_all, world, n, extra=("twopass",))
    late = {k for k in range(n_all) if k % 5 == 3}
    all_refs, all_check = set(), set()
    for a in range(world):
        for b in range(world):
            if a != b:
                s = [k for p, k in zip(got[a]["send_peer"], got[a]["send_kf"]) if p == b]
                r_ = [k for p, k in zip(got[b]["recv_peer"], got[b]["recv_kf"]) if p == a]
                assert s == r_, (a, b, s, r_)
        count = n_all // world
        assert all(a * count <= k < (a + 1) * count for k in got[a]["refs"] + got[a]["check"])
        all_refs.update(got[a]["refs"])
        all_check.update(got[a]["check"])
        recv = set(got[a]["recv_kf"])
        own = set(range(a * count, (a + 1) * count))
        assert recv == set(got[a]["check_nbrs"]) - own  # exactly the maps this rank's checks read from elsewhere
    assert all_refs == late  # pass 2 reconstructs the late keyframes only
    assert all_check == late  # and checks them (everything else was checked in pass 1, against the neighbours it had then)
    assert sum(len(g["recv_kf"]) for g in got) > 0  # their checks read maps reconstructed by other ranks in pass 1
