"""Debug aid: K1 / K2 / K3 / recon of keyframe 0 against the oracle at 1920x1080, N = 7, for a sequence of n_total keyframes
reconstructed in ONE call (how the 2^31-work-item launch limit was found).
usage: python tools/debug/stage_vs_oracle_1080p.py N_TOTAL EXT_POOL(0|1)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sdm_pkg
from pm_oracle import Oracle
pkg = sdm_pkg.load(); oracle = Oracle("strict")
n_total = int(sys.argv[1]); ext = int(sys.argv[2]); n = 7
cam = pkg.synth.HD1080; W, H = cam["W"], cam["H"]
scene = pkg.synth.Scene(cam, 0x5EED0004); Kc = scene.K(); mn, mx = scene.depth_prior()
pool = None
if ext:
    pool = torch.zeros((n_total, H, W, 2), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
eng = pkg.Engine(W, H, n_total, max_neighbours=n, batch_capacity=64, with_pointset=bool(ext), ext_depth_pool=pool.data_ptr() if ext else None)
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
r2, s2 = oracle.intra_check(r1, s1)
g0, _, _ = oracle.gradient_prepass(ims[0])
r3, s3 = oracle.intra_grow(r2, s2, g0)
def cmp(tag, a, b):
    d = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    print(tag, int(d.sum()), "differ of", a.size, flush=True)
eng.search_fuse(refs, nbrs, mn, mx); gr, gs = eng.download_depth(0); cmp("K1 rho", gr, r1)
eng.intra_check(refs); gr, gs = eng.download_depth(0); cmp("K2 rho", gr, r2); cmp("K2 sig", gs, s2)
eng.intra_grow(refs); gr, gs = eng.download_depth(0); cmp("K3 rho", gr, r3); cmp("K3 sig", gs, s3)
eng.recon(refs, nbrs, mn, mx); gr, gs = eng.download_depth(0); cmp("recon rho", gr, r3); cmp("recon sig", gs, s3)
eng.recon(refs[:64], nbrs[:64], mn, mx); gr, gs = eng.download_depth(0); cmp("recon[:64] rho", gr, r3)
