"""What the multi-GPU step costs on the COMPUTE side, measured on one GPU: the workload of a middle rank of an 8-rank job
(own block + input halo resident), once as the plain one-GPU step (K1-K3 over the block, K4+K5) and once as the boundary
all-gather step (boundary keyframes first, packing + gather + fetch of as many maps as really cross ranks, interior
keyframes and their K4+K5, then the boundary keyframes' K4+K5) with the transfer replaced by a device copy.  The difference is everything the sharded step adds except
transfer time: split launches, the copies, the stream hand-overs.  The halo maps are copies of this rank's own boundary
maps (values are realistic, K4's results are not the job's -- this is a timing tool).
usage: python tools/step_overhead.py [--res 480p --kfs 64 --nbrs 20 --world 8 --rank 3 --steps 60]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sdm_pkg  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="480p")
ap.add_argument("--kfs", type=int, default=64)
ap.add_argument("--nbrs", type=int, default=20)
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
pkg = sdm_pkg.load()
shard = pkg.shard
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
wl = bench.Workload(pkg, torch, a.res, a.kfs, a.nbrs, 2.6, a.world, a.rank, 0)
eng, pl = wl.eng, wl.pl
own, nbrs = pl["own_slots"], pl["nbr_slots"]
nb_of = dict(zip(own, nbrs))
boundary, interior = pl["boundary_slots"], pl["interior_slots"]
early, late_slots = pl["check_early_slots"], pl["check_late_slots"]
contrib = shard.contrib_slots(pl)
cc = pl["contrib_count"]
fake_fetch = [(i % cc, s) for i, (_, s) in enumerate(shard.contrib_fetch_list(pl))]
print("rank %d of %d: %d own keyframes (%d boundary, %d interior), %d maps contributed, %d fetched" % (
    a.rank, a.world, len(own), len(boundary), len(interior), cc, len(fake_fetch)))


def plain():
    eng.recon(own, nbrs, wl.min_d, wl.max_d)
    eng.inter_check_pointset(own, nbrs, commit=False)


def split():
    eng.recon(boundary, [nb_of[k] for k in boundary], wl.min_d, wl.max_d)
    eng.allgather_begin(cc)
    eng.allgather_piece(contrib)
    eng.recon(interior, [nb_of[k] for k in interior], wl.min_d, wl.max_d)
    eng.inter_check_pointset(early, [nb_of[k] for k in early], commit=False)
    eng.allgather_finish(fake_fetch)
    eng.inter_check_pointset(late_slots, [nb_of[k] for k in late_slots], commit=False)


def late():  # the boundary all-gather after an UNSPLIT reconstruction (exchange="allgather_late")
    eng.recon(own, nbrs, wl.min_d, wl.max_d)
    eng.allgather_begin(cc)
    eng.allgather_piece(contrib)
    eng.inter_check_pointset(early, [nb_of[k] for k in early], commit=False)
    eng.allgather_finish(fake_fetch)
    eng.inter_check_pointset(late_slots, [nb_of[k] for k in late_slots], commit=False)


split()  # fills the halo slots
torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    plain()
torch.cuda.synchronize()
res = {"plain": [], "split": [], "late": []}
for _ in range(a.rounds):
    for name, fn in (("plain", plain), ("split", split), ("late", late)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / a.steps * 1e3)
eng.enable_timing(True)
for name, fn in (("plain", plain), ("split", split), ("late", late)):
    eng.get_timing(reset=True)
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    tm = eng.get_timing(reset=True)
    print("%s stage ms per step: %s" % (name, ", ".join("%s %.4f (%d launches)" % (k, v[0] / a.steps, v[1] // a.steps)
                                                        for k, v in tm.items() if v[1])))
eng.enable_timing(False)
for name in res:
    v = sorted(res[name])
    print("%s %s: median %.4f ms per step (min %.4f, max %.4f)" % (a.res, name, v[len(v) // 2], v[0], v[-1]))
p, s = sorted(res["plain"])[a.rounds // 2], sorted(res["split"])[a.rounds // 2]
print("compute-side ceiling of the weak-scaling efficiency (transfer fully hidden): %.3f split, %.3f late" % (
    p / s, p / sorted(res["late"])[a.rounds // 2]))
wl.close()
