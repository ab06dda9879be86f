"""GPU: random sequences of engine calls against a host-side model driven by the oracle.

The engine keeps per-slot state that decides which kernels run (pixel lists and the lambdaG they were built with,
"pipeline map" and "plane is zero outside the list" flags, cached table sets, pose epochs).  Each sequence mixes
uploads, pose and parameter changes, the individual stages, the batched forms and the fused call on a small scene;
after every call the affected keyframes' depth map, checked plane and point set must equal the model's bit for bit."""
import os

import numpy as np
import pytest

from common import Sequence, assert_bit_equal

pytestmark = pytest.mark.gpu


class Model:
    """what the reference's KeyFrame members would hold after the same calls (oracle arithmetic)"""

    def __init__(self, oracle, W, H, n_kf, K):
        self.o, self.W, self.H, self.n_kf, self.K = oracle, W, H, n_kf, K
        z = lambda *s: np.zeros(s, np.float32)
        self.im, self.Tcw, self.der = {}, {}, {}
        self.rho = {k: z(H, W) for k in range(n_kf)}
        self.sig = {k: z(H, W) for k in range(n_kf)}
        self.chk = {k: z(H, W) for k in range(n_kf)}
        self.xyz = {k: z(H, 3 * W) for k in range(n_kf)}
        self.has_chk = {k: False for k in range(n_kf)}
        self.has_depth = {k: False for k in range(n_kf)}  # kf->semidense_flag_ (PM.cc:244, gate at :292-298)

    def kf(self, k):
        g, th, s = self.der[k]
        return self.o.keyframe(self.im[k], g, th, s, self.K, self.Tcw[k])

    def upload_image(self, k, im, Tcw):
        self.im[k], self.Tcw[k] = im, Tcw
        self.der[k] = self.o.gradient_prepass(im)
        for m in (self.rho, self.sig, self.chk):
            m[k] = np.zeros((self.H, self.W), np.float32)
        self.xyz[k] = np.zeros((self.H, 3 * self.W), np.float32)
        self.has_chk[k] = False
        self.has_depth[k] = False


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("SDM_FUZZ_FIRST", "1")), int(os.environ.get("SDM_FUZZ_FIRST", "1")) +
                                             int(os.environ.get("SDM_FUZZ_SEEDS", "8")))))  # env: deeper one-off runs
@pytest.mark.parametrize("overlap", [False, True])
def test_random_call_sequences(pkg, oracle, gpu_ok, seed, overlap):
    rng = np.random.default_rng(7000 + seed)
    W, H, n_kf, n = 96, 72, 8, 5
    seq = Sequence(pkg, oracle, W, H, n_kf, 0x5EED0F00 + seed)  # consistent geometry: the checks keep many pixels
    K = seq.K
    eng = pkg.Engine(W, H, n_kf, max_neighbours=n, with_pointset=True)
    eng.set_ingest_overlap(overlap)  # batch uploads of >= 5 keyframes then run next to whatever does not use their slots
    m = Model(oracle, W, H, n_kf, K)
    lam = 8.0
    oracle.params.lambdaG = lam
    kept = refused = 0
    try:
        for k in range(n_kf):
            eng.upload_image(k, seq.im[k], K, seq.Tcw[k])
            m.upload_image(k, seq.im[k], seq.Tcw[k])
        mind, maxd = seq.min_depth, seq.max_depth
        for step in range(160):
            refs = sorted(rng.choice(n_kf, int(rng.integers(1, 5)), replace=False).tolist())
            if rng.random() < 0.8:
                nbrs = [seq.neighbours(k, n) for k in refs]
            else:
                nbrs = [[int(j) for j in rng.permutation([j for j in range(n_kf) if j != k])[:n]] for k in refs]
            op = rng.choice(["recon"] * 6 + ["fused"] * 4 + ["inter"] * 3 + ["inter_commit", "search_fuse", "intra_check",
                             "intra_grow", "pointset0", "pointset1", "pointset1", "upload_depth", "assume", "set_pose",
                             "lambda", "lambda", "reupload", "recon+batch"])
            touched = list(refs)
            if op in ("recon+batch", "batch"):
                # a reconstruction is queued and, without waiting for it, a batch of 5 .. 8 keyframes (other images: the
                # neighbour's) is uploaded into slots that may be the ones it reads or writes: the engine must order them
                if op == "recon+batch":
                    eng.recon(refs, nbrs, mind, maxd)
                    for k, nb in zip(refs, nbrs):
                        m.rho[k], m.sig[k], _ = oracle.semi_dense_recon(m.kf(k), [m.kf(j) for j in nb], None, mind, maxd)
                        m.has_depth[k] = True
                bs = sorted(rng.choice(n_kf, int(rng.integers(5, n_kf + 1)), replace=False).tolist())
                # (the same views a little brighter or darker: new records, lists and maps, the geometry stays consistent)
                d = int(rng.integers(-3, 4))
                ims = [np.clip(seq.im[k].astype(np.int32) + d, 0, 255).astype(np.uint8) for k in bs]
                eng.upload_images_batch(bs, ims, K, [seq.Tcw[k] for k in bs])
                for k, im in zip(bs, ims):
                    m.upload_image(k, im, seq.Tcw[k])
                # ... and the new keyframes are reconstructed at once (queued behind the upload; keeps the sequence's
                # inter-keyframe checks alive: a fresh keyframe has no depth map)
                bn = [seq.neighbours(k, n) for k in bs]
                eng.recon(bs, bn, mind, maxd)
                for k, nb in zip(bs, bn):
                    m.rho[k], m.sig[k], _ = oracle.semi_dense_recon(m.kf(k), [m.kf(j) for j in nb], None, mind, maxd)
                    m.has_depth[k] = True
                touched = sorted(set(refs if op == "recon+batch" else []) | set(bs))
            elif op == "recon":
                eng.recon(refs, nbrs, mind, maxd)
                for k, nb in zip(refs, nbrs):
                    m.rho[k], m.sig[k], _ = oracle.semi_dense_recon(m.kf(k), [m.kf(j) for j in nb], None, mind, maxd)
                    m.has_depth[k] = True
            elif op == "search_fuse":
                eng.search_fuse(refs, nbrs, mind, maxd)
                for k, nb in zip(refs, nbrs):
                    m.rho[k], m.sig[k], _ = oracle.recon_search_fuse(m.kf(k), [m.kf(j) for j in nb], None, mind, maxd)
                    m.has_depth[k] = True
            elif op == "intra_check":
                eng.intra_check(refs)
                for k in refs:
                    m.rho[k], m.sig[k] = oracle.intra_check(m.rho[k], m.sig[k])
            elif op == "intra_grow":
                eng.intra_grow(refs)
                for k in refs:
                    m.rho[k], m.sig[k] = oracle.intra_grow(m.rho[k], m.sig[k], m.der[k][0])
            elif op in ("inter", "inter_commit", "fused"):
                commit = op == "inter_commit"
                if not all(m.has_depth[k] for k in refs) or not all(m.has_depth[j] for nb in nbrs for j in nb):
                    # the reference's gate (PM.cc:292-298): a keyframe or neighbour without a depth map is a caller
                    # error in the C ABI -- SDM_ESTATE, and nothing changes
                    with pytest.raises(pkg.SdmError) as ei:
                        (eng.inter_check_pointset if op == "fused" else eng.inter_check)(refs, nbrs)
                    assert ei.value.code == 4
                    refused += 1
                    continue
                if op == "fused":
                    eng.inter_check_pointset(refs, nbrs)
                else:
                    eng.inter_check(refs, nbrs, commit=commit)
                new = {}
                for k, nb in zip(refs, nbrs):  # every reference is checked against the maps as they were before the call
                    new[k] = oracle.inter_check(m.kf(k), m.rho[k], [m.kf(j) for j in nb], [m.rho[j] for j in nb],
                                                [m.sig[j] for j in nb])
                for k in refs:
                    m.chk[k] = new[k]
                    m.has_chk[k] = True
                    if commit:
                        m.rho[k] = new[k].copy()
                    if op == "fused":
                        m.xyz[k] = oracle.pointset(m.kf(k), m.chk[k])
            elif op in ("pointset0", "pointset1"):
                src = 1 if op == "pointset1" else 0
                if src == 1:
                    refs = [k for k in refs if m.has_chk[k]]
                    touched = list(refs)
                    if not refs:
                        continue
                eng.pointset(refs, source=src)
                for k in refs:
                    m.xyz[k] = oracle.pointset(m.kf(k), m.chk[k] if src else m.rho[k])
            elif op == "upload_depth":
                k = refs[0]
                touched = [k]
                r = np.where(rng.random((H, W)) < 0.3, rng.uniform(0.5, 1.5, (H, W)), 0).astype(np.float32)
                s = np.where(r > 0, rng.uniform(0.01, 0.2, (H, W)), 0).astype(np.float32)
                r[:2] = r[-2:] = 0
                r[:, :2] = r[:, -2:] = 0
                eng.upload_depth(k, r, s)
                m.rho[k], m.sig[k] = r, s
                m.has_depth[k] = True
            elif op == "assume":
                # legitimate only for maps ({rho, sigma}) that are zero outside the current list
                def zero_outside(k):
                    inside = np.zeros((H, W), bool)
                    inside[2:-2, 2:-2] = m.der[k][0][2:-2, 2:-2] >= lam
                    return not m.rho[k][~inside].any() and not m.sig[k][~inside].any()
                ok = [k for k in refs if zero_outside(k)]
                touched = ok
                if ok:
                    eng.assume_pipeline_maps(ok)
                    for k in ok:
                        m.has_depth[k] = True
            elif op == "set_pose":
                k = refs[0]
                touched = [k]
                T = seq.Tcw[k].copy()
                T[:, 3] += rng.normal(0, 1e-4, 3).astype(np.float32)
                eng.set_pose(k, T)
                m.Tcw[k] = T
            elif op == "lambda":
                lam = float(rng.choice([8.0, 12.0, 5.0]))
                eng.set_params(lambdaG=lam)
                oracle.params.lambdaG = lam
                touched = []
            elif op == "reupload":
                k = refs[0]
                touched = [k]
                eng.upload_image(k, seq.im[k], K, seq.Tcw[k])
                m.upload_image(k, seq.im[k], seq.Tcw[k])
            for k in touched:
                what = "seed %d step %d %s kf %d" % (seed, step, op, k)
                gr, gs = eng.download_depth(k)
                assert_bit_equal(gr, m.rho[k], what + " rho")
                assert_bit_equal(gs, m.sig[k], what + " sigma")
                if m.has_chk[k]:
                    assert_bit_equal(eng.download_checked(k), m.chk[k], what + " checked")
                    kept += int((m.chk[k] > 1e-6).sum())
                assert_bit_equal(eng.download_pointset(k), m.xyz[k], what + " xyz")
        assert kept > 2000, "the sequences must keep inter-keyframe-checked pixels alive"
    finally:
        oracle.params.lambdaG = 8.0
        eng.close()
