"""Generates the committed golden fixtures from the CPU oracle (oracle/pm_oracle.c, strict build:
gcc -O2 -ffp-contract=off).  PARITY UNPINNED: the reference holds no vectors for this path and
cannot be built here, so these fixtures pin the ORACLE, not the reference.

Each fixture = the synthetic images (committed, because the float64 `sin` of the scene generator is
not guaranteed bit-reproducible across libm/GPU versions), the poses/intrinsics, and every stage
output of the oracle for every keyframe:  search+fuse (K1), intra check+grow (K3), inter-keyframe
check with snapshot semantics (K4) as float32 bit patterns; the gradient inputs and the point set
(K5) as SHA-256 digests.

    python tests/golden/make_golden.py [name ...]   # rewrites tests/golden/*.npz (all, or the named ones)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

FIXTURES = [
    # name, W, H, n_kf, n_nbr, seed, disparity_px, scene options
    ("plane_64x48_n7", 64, 48, 8, 7, 0x5EED0101, 2.6, {}),
    ("plane_160x120_n7", 160, 120, 8, 7, 0x5EED0102, 2.6, {}),
    ("plane_96x80_n20", 96, 80, 21, 20, 0x5EED0103, 1.5, {}),
    # App. D's second plane strip (a depth discontinuity with occluding edges) and keyframes rolled by up to +-5 degrees:
    # pairs rotated against each other by up to 10 degrees, with the matching median rotations (PM.cc:170-179)
    ("strip_roll_160x120_n7", 160, 120, 8, 7, 0x5EED0104, 3.0, {"strip": True, "roll_deg": 5.0}),
]


def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def main():
    import sdm_pkg
    from pm_oracle import Oracle
    from common import Sequence, oracle_inter, oracle_pipeline
    pkg = sdm_pkg.load()
    oracle = Oracle("strict")
    only = sys.argv[1:]
    for name, W, H, n_kf, n, seed, disp, opts in FIXTURES:
        if only and name not in only:
            continue
        seq = Sequence(pkg, oracle, W, H, n_kf, seed, disparity_px=disp, **opts)
        maps = oracle_pipeline(oracle, seq, n)
        chk, xyz = oracle_inter(oracle, seq, n, maps)
        out = dict(
            meta=np.array([W, H, n_kf, n, seed], dtype=np.int64),
            disparity_px=np.float64(disp),
            K=seq.K, Tcw=np.stack(seq.Tcw), im=np.stack(seq.im),
            min_depth=np.float32(seq.min_depth), max_depth=np.float32(seq.max_depth),
            nbrs=np.array([seq.neighbours(k, n) for k in range(n_kf)], dtype=np.int32),
            # derived inputs and the point set are pinned by SHA-256 of their float32 bytes
            grad_sha=np.array([sha(g) for g in seq.grad]), theta_sha=np.array([sha(t) for t in seq.theta]),
            istd=np.array(seq.istd, dtype=np.float32),
            k1_rho=np.stack([maps["k1_rho"][k] for k in range(n_kf)]).view(np.uint32),
            k1_sigma=np.stack([maps["k1_sigma"][k] for k in range(n_kf)]).view(np.uint32),
            rho=np.stack([maps["rho"][k] for k in range(n_kf)]).view(np.uint32),
            sigma=np.stack([maps["sigma"][k] for k in range(n_kf)]).view(np.uint32),
            chk=np.stack([chk[k] for k in range(n_kf)]).view(np.uint32),
            xyz_sha=np.array([sha(xyz[k]) for k in range(n_kf)]),
            searches=np.array([maps["stats"][k]["searches"] for k in range(n_kf)], dtype=np.int64),
            candidates=np.array([maps["stats"][k]["candidates"] for k in range(n_kf)], dtype=np.int64),
            fused=np.array([maps["stats"][k]["fused"] for k in range(n_kf)], dtype=np.int64),
        )
        if opts:
            out["strip"] = np.int64(1 if opts.get("strip") else 0)
            out["roll_deg"] = np.float64(opts.get("roll_deg", 1.0))
            out["rot"] = seq.rots(range(n_kf), n)
            out["gt_rho"] = np.stack(seq.gt).astype(np.float32)
            out["fg"] = np.stack([seq.fg[k] for k in range(n_kf)])
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        kept = int(sum((chk[k] > 1e-6).sum() for k in range(n_kf)))
        print("%s: %d KB, fused %d, kept after inter-check %d" % (name, os.path.getsize(path) // 1024,
                                                                  int(out["fused"].sum()), kept))


if __name__ == "__main__":
    main()
