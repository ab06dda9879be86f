import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sdm_pkg
from pm_oracle import Oracle
pkg = sdm_pkg.load(); oracle = Oracle("strict")
n_total = int(sys.argv[1]); n = 7
cam = pkg.synth.HD1080; W, H = cam["W"], cam["H"]
scene = pkg.synth.Scene(cam, 0x5EED0004); Kc = scene.K(); mn, mx = scene.depth_prior()
eng = pkg.Engine(W, H, n_total, max_neighbours=n, batch_capacity=64, with_pointset=False)
refs = list(range(n_total)); nbrs = [scene.neighbours(k, n_total, n) for k in refs]
keep = {0} | set(nbrs[0]); ims = {}
for k in refs:
    im, _ = scene.render(k, device="cuda"); torch.cuda.synchronize()
    eng.upload_image_device(k, im.data_ptr(), Kc, scene.Tcw(k))
    if k in keep: ims[k] = im.cpu().numpy()
kf = {}
for j in keep:
    g, t, s_ = oracle.gradient_prepass(ims[j]); kf[j] = oracle.keyframe(ims[j], g, t, s_, Kc, scene.Tcw(j))
r1, s1, st = oracle.recon_search_fuse(kf[0], [kf[j] for j in nbrs[0]], None, mn, mx)
def cmp(tag, a, b):
    d = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    print(tag, int(d.sum()), "differ of", a.size, flush=True)
for label, call in (("all refs", lambda: eng.search_fuse(refs, nbrs, mn, mx)), ("kf0 only", lambda: eng.search_fuse([0], [nbrs[0]], mn, mx)),
                    ("first 64", lambda: eng.search_fuse(refs[:64], nbrs[:64], mn, mx))):
    call(); gr, gs = eng.download_depth(0)
    cmp("K1 rho kf0 [%s]" % label, gr, r1); cmp("K1 sig kf0 [%s]" % label, gs, s1)
eng.enable_stats(True); eng.get_stats(reset=True); eng.search_fuse(refs, nbrs, mn, mx); print(eng.get_stats()); eng.enable_stats(False)
