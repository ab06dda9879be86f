"""Debug aid for the round-3 finding "K1 over 2048 keyframes of 1920x1080 in ONE launch returns wrong maps" (DESIGN.md §7):
runs each stage of SemiDenseRecon over n_total keyframes twice -- launches sliced below 2^30 work-items (the shipped form)
and unsliced (SDM_MAX_DISPATCH_LOG2=40) -- and lists the keyframes whose maps differ, stage by stage.
usage: python tools/debug/unsliced_vs_sliced.py [N_TOTAL=2048]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sdm_pkg  # noqa: E402

pkg = sdm_pkg.load()
n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = 7
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
print("uploaded", n_total, "keyframes; active pixels of kf 0:", eng.active_count(0), flush=True)


def sums():
    eng.synchronize()
    torch.cuda.synchronize()
    out = []
    for k0 in range(0, n_total, 64):  # per-keyframe sums of the bit patterns (int64: no overflow)
        out.append(pool[k0:k0 + 64].view(torch.int32).to(torch.int64).sum(dim=(1, 2, 3)))
    return torch.cat(out).cpu()


def stage(name, fn):
    res = {}
    for mode, lg, sync in (("sliced", "30", "0"), ("unsliced", "40", "0"), ("unsliced+sync", "40", "1"), ("sliced again", "30", "0")):
        os.environ["SDM_MAX_DISPATCH_LOG2"] = lg
        os.environ["SDM_DEBUG_SYNC_K1"] = sync
        fn()
        res[mode] = sums()
    for mode in ("unsliced", "unsliced+sync", "sliced again"):
        bad = (res[mode] != res["sliced"]).nonzero().flatten().tolist()
        print("%-12s %-12s vs sliced: %d keyframes differ%s" % (name, mode, len(bad), (": " + str(bad[:40])) if bad else ""),
              flush=True)
    return res


stage("K1", lambda: eng.search_fuse(refs, nbrs, mn, mx))
stage("K1+K2+K3", lambda: eng.recon(refs, nbrs, mn, mx))


def k1_then_k2():  # the same two stages as separate calls, nothing waits in between
    eng.search_fuse(refs, nbrs, mn, mx)
    eng.intra_check(refs)


stage("K1, K2 calls", k1_then_k2)
for n_part in (1536, 1200, 1100, 1024):
    stage("recon %d" % n_part, lambda: eng.recon(refs[:n_part], nbrs[:n_part], mn, mx))
eng.close()
