"""GPU: the sharded (multi-rank) hot path gives the single-rank results bit for bit.

Ranks are separate processes that all use GPU 0 (a one-GPU box) with a gloo group and the
host-staged exchange of shard.py; the kernels, the plan, the boundary-first order and the
data that crosses ranks are exactly those of the RCCL run (bench.py --gpus N), only the transport
differs.  Covers SURVEY.md §8e on hardware as far as one GPU allows."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from common import assert_bit_equal
import shard_worker

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


CAMPX = shard_worker.CAM["W"] * shard_worker.CAM["H"]


def _single_rank(pkg, n_total, n_nbr):
    cam = shard_worker.CAM
    scene = pkg.synth.Scene(cam, shard_worker.SEED)
    eng = pkg.Engine(cam["W"], cam["H"], n_total, max_neighbours=n_nbr)
    for k in range(n_total):
        im, _ = scene.render(k, device="cuda")
        torch.cuda.synchronize()
        eng.upload_image_device(k, im.data_ptr(), scene.K(), scene.Tcw(k))
    refs = list(range(n_total))
    nbrs = [scene.neighbours(k, n_total, n_nbr) for k in refs]
    min_d, max_d = scene.depth_prior()
    eng.recon(refs, nbrs, min_d, max_d)
    eng.inter_check(refs, nbrs, commit=False)
    eng.pointset(refs, source=1)
    res = {k: (eng.download_depth(k), eng.download_checked(k), eng.download_pointset(k)) for k in refs}
    eng.close()
    return res


@pytest.mark.parametrize("world,exchange", [(2, "halo"), (2, "allgather_full"), (3, "halo"), (3, "allgather"), (2, "allgather_late"),
                                            (2, "halo+compact"), (3, "allgather+compact")])
def test_sharded_equals_single_rank(pkg, gpu_ok, tmp_path, world, exchange):
    """+compact: the maps cross the PROCESS boundary in the compact wire format (the engine's pack / scatter kernels on
    both sides, gloo in between) -- the format's first trip between two processes with real device lists"""
    n_total, n_nbr = 12, 4
    exchange, _, wire = exchange.partition("+")
    want = _single_rank(pkg, n_total, n_nbr)
    assert sum(int((c > 1e-6).sum()) for (_, c, _) in want.values()) > 1000, "the scene must produce checked depth"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "shard_worker.py"), str(tmp_path), exchange, str(n_total), str(n_nbr)] + ([wire] if wire else [])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    seen, crossed = set(), 0
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        crossed += len(z["recv"])
        assert int(z["refused"][0]) == 0
        if wire:
            assert 0 < int(z["entries"][0]) < CAMPX // 2, "the compact format was in use"
        for k in z["own"].tolist():
            (rho, sig), chk, xyz = want[k]
            assert_bit_equal(z["rho%d" % k], rho, "rho kf %d" % k)
            assert_bit_equal(z["sig%d" % k], sig, "sigma kf %d" % k)
            assert_bit_equal(z["chk%d" % k], chk, "checked rho kf %d" % k)
            assert_bit_equal(z["xyz%d" % k], xyz, "xyz kf %d" % k)
            seen.add(k)
    assert seen == set(range(n_total))
    assert crossed > 0, "the inter-keyframe check must have read maps produced by another rank"
