"""Cost of re-staging the per-call tables (sdm_engine.hip: TableSet cache) and of splitting the step into the
boundary / interior / whole-block calls of the multi-GPU halo schedule, measured on one GPU."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import sdm_pkg
pkg = sdm_pkg.load(); synth = pkg.synth
n_kf, N = 64, 20
cam = synth.TUM1; scene = synth.Scene(cam, 0x5EED0002)
eng = pkg.Engine(cam["W"], cam["H"], n_kf, max_neighbours=N, with_pointset=True)
for k in range(n_kf):
    im, _ = scene.render(k, device="cuda"); torch.cuda.synchronize()
    eng.upload_image_device(k, im.data_ptr(), scene.K(), scene.Tcw(k))
mn, mx = scene.depth_prior()
nb = [scene.neighbours(k, n_kf, N) for k in range(n_kf)]
allr = list(range(n_kf))
eng.recon(allr, nb, mn, mx); eng.synchronize()
A = allr[:20]; B = allr[20:]
def t(fn, reps=200):
    fn(); eng.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    eng.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
same = t(lambda: (eng.inter_check(A, [nb[k] for k in A]), eng.inter_check(A, [nb[k] for k in A])))
alt = t(lambda: (eng.inter_check(A, [nb[k] for k in A]), eng.inter_check(B, [nb[k] for k in B])))
onlyB = t(lambda: (eng.inter_check(B, [nb[k] for k in B]), eng.inter_check(B, [nb[k] for k in B])))
print("A,A %.3f ms   B,B %.3f ms   A,B alternating %.3f ms  -> staging cost per call ~ %.3f ms" % (same, onlyB, alt, (alt - (same + onlyB) / 2) / 2))
bnd = allr[:10] + allr[-10:]
inr = allr[10:-10]
whole = t(lambda: (eng.recon(allr, nb, mn, mx), eng.inter_check_pointset(allr, nb)), 50)
split = t(lambda: (eng.recon(bnd, [nb[k] for k in bnd], mn, mx), eng.recon(inr, [nb[k] for k in inr], mn, mx),
                   eng.inter_check_pointset(allr, nb)), 50)
print("step as one recon call %.3f ms; as boundary(20) + interior(44) recon calls %.3f ms" % (whole, split))
eng.close()
