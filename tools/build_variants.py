"""Builds A/B variants of the engine for kernel experiments: one libsdm_hip_<tag>.so per set of -D flags under
orb-slam-free-space-carving_amd/lib/variants/ (git-ignored, travels to the GPU box).  SDM_LIB_PATH selects one at run time
(binding.lib_path).  usage: python tools/build_variants.py tag=-DFLAG[,-DFLAG2] ...   e.g.  opt00=-DSDM_K1_OPT=0x00"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "orb-slam-free-space-carving_amd")
sys.path.insert(0, PKG)
import build as b  # noqa: E402

out_dir = os.path.join(b.LIB, "variants")
os.makedirs(out_dir, exist_ok=True)


def one(spec):
    tag, flags = spec.split("=", 1)
    out = os.path.join(out_dir, "libsdm_hip_%s.so" % tag)
    cmd = ["hipcc", "--offload-arch=" + b.ARCH] + b.COMMON + flags.split(",") + [os.path.join(PKG, "csrc", "sdm_engine.hip"), "-o", out]
    subprocess.check_call(cmd)
    return out


with ThreadPoolExecutor(4) as ex:
    for o in ex.map(one, sys.argv[1:]):
        print(o)
