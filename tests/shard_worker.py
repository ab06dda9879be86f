"""Worker for tests/test_gpu_shard.py: one rank of the sharded hot path.  Launched by
torch.distributed.run with every rank on GPU 0 and a gloo group (host-staged exchange, shard.py),
so the multi-rank control flow -- block partition, input halo, boundary-first reconstruction, exchange
of {rho,sigma} maps, inter-keyframe check against received maps -- runs on a one-GPU box.

Slots are LOCAL (own block + input halo, shard.plan): the engine of a rank holds len(inputs) keyframes.

usage: shard_worker.py OUT_DIR EXCHANGE N_TOTAL N_NBR [compact]
(compact: the maps cross the process boundary in the compact wire format -- packed and scattered by the engine's own
kernels, sdm_compact_pack_host / _unpack_host, staged through host memory; the entries per map are agreed over the group)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdm_pkg  # noqa: E402

CAM = dict(W=160, H=120, fx=129.3266, fy=129.1173, cx=79.6608, cy=63.8285)  # TUM1 / 4
SEED = 0x5EED0E01


def main():
    out_dir, exchange, n_total, n_nbr = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    pkg = sdm_pkg.load()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    scene = pkg.synth.Scene(CAM, SEED)
    W, H = CAM["W"], CAM["H"]
    pl = pkg.shard.plan(n_total, world, rank, n_nbr, scene.neighbours)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    slot = pl["slot"]
    pool = torch.zeros((pl["n_slots"], H, W, 2), dtype=torch.float32, device="cuda")
    eng = pkg.Engine(W, H, pl["n_slots"], max_neighbours=n_nbr, device=0, batch_capacity=pl["n_slots"],
                     with_pointset=True, ext_depth_pool=pool.data_ptr(), stream=stream.cuda_stream)
    for k in pl["inputs"]:  # own block + input halo only
        im, _ = scene.render(k, device="cuda")
        torch.cuda.synchronize()
        eng.upload_image_device(slot[k], im.data_ptr(), scene.K(), scene.Tcw(k))
    min_d, max_d = scene.depth_prior()
    compact = len(sys.argv) > 5 and sys.argv[5] == "compact"
    entries = pkg.shard.agree_compact_wire(eng, pl) if compact else 0
    for _ in range(2):  # twice: the second pass must not depend on state left by the first
        pkg.shard.pipeline_step(eng, pool, pl, min_d, max_d, exchange, transport="torch")
    torch.cuda.synchronize()
    out = {"entries": np.array([entries]), "refused": np.array([getattr(eng, "staged_refused", 0)]),
           "own": np.array(pl["own"]), "recv": np.array(sorted(j for v in pl["recv"].values() for j in v))}
    for k in pl["own"]:
        r, s = eng.download_depth(slot[k])
        out["rho%d" % k], out["sig%d" % k] = r, s
        out["chk%d" % k] = eng.download_checked(slot[k])
        out["xyz%d" % k] = eng.download_pointset(slot[k])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
