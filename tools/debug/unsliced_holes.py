"""Third debug aid for the 2048-keyframe 1080p finding: WHICH list positions does an unsliced K1 launch leave unwritten?
The pool is filled with a sentinel, K1 runs (pipeline maps: no zero-fill), and the listed pixels that still hold the
sentinel are mapped back to their list positions.  usage: python tools/debug/unsliced_holes.py [N_TOTAL=2048]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
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
os.environ["SDM_MAX_DISPATCH_LOG2"] = "30"
eng.search_fuse(refs, nbrs, mn, mx)  # the maps become pipeline maps: later K1 launches do not zero-fill
eng.synchronize()
for lg in ("30", "40"):
    for n_part in (n_total, n_total * 3 // 4):
        pool.fill_(7.0)
        torch.cuda.synchronize()
        os.environ["SDM_MAX_DISPATCH_LOG2"] = lg
        eng.enable_stats(True)
        eng.get_stats(reset=True)
        eng.search_fuse(refs[:n_part], nbrs[:n_part], mn, mx)
        eng.synchronize()
        st = eng.get_stats()
        eng.enable_stats(False)
        torch.cuda.synchronize()
        print("lg %s, %d keyframes in one call: searches counted %d" % (lg, n_part, st["searches"]), flush=True)
        for k in (0, 1, n_part // 2, n_part - 1):
            lst, _ = eng.active_list(k)
            ys, xs = (lst >> 16).astype(np.int64), (lst & 0xFFFF).astype(np.int64)
            m = pool[k, :, :, 1].cpu().numpy()[ys, xs] == 7.0  # sigma still the sentinel: the pixel was not written
            holes = np.nonzero(m)[0]
            if holes.size == 0:
                print("   kf %4d: %d listed, all written" % (k, lst.size), flush=True)
                continue
            ch = holes // 64
            runs = np.nonzero(np.diff(ch) > 1)[0].size + 1
            print("   kf %4d: %d listed, %d unwritten; list positions %d..%d, chunks %d..%d in %d run(s); chunk %% 8 histogram %s" % (
                k, lst.size, holes.size, holes[0], holes[-1], ch[0], ch[-1], runs,
                np.bincount(np.unique(ch) % 8, minlength=8).tolist()), flush=True)
eng.close()
