"""Loading of tests/golden/*.npz (see tests/golden/make_golden.py)."""
import glob
import hashlib
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixture_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    for k in ("k1_rho", "k1_sigma", "rho", "sigma", "chk"):
        d[k] = d[k].view(np.float32)
    d["W"], d["H"], d["n_kf"], d["n"], d["seed"] = [int(v) for v in d["meta"]]
    return d


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def sequence_from(pkg, oracle, g):
    from common import Sequence
    return Sequence(pkg, oracle, g["W"], g["H"], g["n_kf"], g["seed"], disparity_px=float(g["disparity_px"]),
                    images=g["im"], **scene_options(g))


def scene_options(g):
    return dict(strip=bool(int(g["strip"])), roll_deg=float(g["roll_deg"])) if "roll_deg" in g else {}


def rots(g):
    """[n_kf, n] median in-plane rotations of the fixture's pairs (PM.cc:170-179), or None (App. D: 0)"""
    return g["rot"] if "rot" in g else None
