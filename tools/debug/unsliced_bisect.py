"""Second debug aid for the 2048-keyframe 1080p finding (see unsliced_vs_sliced.py): which stage is sensitive to the
dispatch-size knob, and what exactly differs.  usage: python tools/debug/unsliced_bisect.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import sdm_pkg  # noqa: E402

pkg = sdm_pkg.load()
n_total, n = 2048, 7
cam = pkg.synth.HD1080
W, H = cam["W"], cam["H"]
scene = pkg.synth.Scene(cam, 0x5EED0004)
Kc = scene.K()
mn, mx = scene.depth_prior()
pool = torch.zeros((n_total, H, W, 2), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
eng = pkg.Engine(W, H, n_total, max_neighbours=n, batch_capacity=64, with_pointset=False, ext_depth_pool=pool.data_ptr())
refs = list(range(n_total))
nbrs = [scene.neighbours(k, n_total, n) for k in refs]
for k in refs:
    im, _ = scene.render(k, device="cuda")
    torch.cuda.synchronize()
    eng.upload_image_device(k, im.data_ptr(), Kc, scene.Tcw(k))
lists0 = {k: eng.active_list(k) for k in (0, 1, 1000, 2047)}
print("uploaded", flush=True)


def run(lg1, lg2):
    os.environ["SDM_MAX_DISPATCH_LOG2"] = lg1
    eng.search_fuse(refs, nbrs, mn, mx)
    eng.synchronize()
    torch.cuda.synchronize()
    k1 = pool.clone()
    torch.cuda.synchronize()  # the copy runs on torch's stream: finished before the engine's stream touches the pool again
    os.environ["SDM_MAX_DISPATCH_LOG2"] = lg2
    eng.intra_check(refs)
    eng.synchronize()
    torch.cuda.synchronize()
    k2 = pool.clone()
    torch.cuda.synchronize()
    return k1, k2


def diff(tag, a, b):
    bad = []
    for k0 in range(0, n_total, 64):
        d = (a[k0:k0 + 64].view(torch.int32) != b[k0:k0 + 64].view(torch.int32)).sum(dim=(1, 2, 3))
        bad += [(k0 + int(i), int(d[i])) for i in d.nonzero().flatten().tolist()]
    print("%-34s keyframes that differ: %d %s" % (tag, len(bad), bad[:6]), flush=True)
    return bad


ref_k1, ref_k2 = run("30", "30")
for lg1, lg2 in (("40", "30"), ("30", "40"), ("40", "40")):
    k1, k2 = run(lg1, lg2)
    diff("K1(lg %s) vs sliced" % lg1, k1, ref_k1)
    bad = diff("K1(lg %s) + K2(lg %s) vs sliced" % (lg1, lg2), k2, ref_k2)
    if bad:
        k = bad[0][0]
        d = (k2[k].view(torch.int32) != ref_k2[k].view(torch.int32)).any(dim=2).nonzero()
        y, x = [int(v) for v in d[0]]
        lst = lists0.get(k, eng.active_list(k))[0]
        print("   kf %d first at (y %d, x %d): got %s want %s; K1 there %s; listed %s; rows touched %d..%d" % (
            k, y, x, k2[k, y, x].tolist(), ref_k2[k, y, x].tolist(), ref_k1[k, y, x].tolist(),
            bool(((lst >> 16) == y).any() and (lst[(lst >> 16) == y] & 0xFFFF == x).any()), int(d[:, 0].min()), int(d[:, 0].max())),
            flush=True)
    del k1, k2
for k, (l0, h0) in lists0.items():
    l1, h1 = eng.active_list(k)
    print("list kf %d unchanged: %s" % (k, bool(l0.size == l1.size and (l0 == l1).all() and h0 == h1)), flush=True)
eng.close()
