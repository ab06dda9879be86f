"""Builds the gfx950 shared libraries in-tree (they travel to the GPU box with the snapshot).

  lib/libsdm_hip.so  HIP kernels + C ABI (include/sdm_c.h)
  lib/libsdm_pm.so   C++ ProbabilityMapping class mirror (include/sdm/ProbabilityMapping.h)

hipcc cross-compiles without a GPU.  -ffp-contract=off is load-bearing: the reference's float /
double promotion pattern must not be fused into FMAs.  -fno-slp-vectorize
keeps hipcc from packing adjacent scalar f32 ops into v_pk_*_f32, which are slower than the two
scalar instructions on gfx950 (measured: K1 1.99 -> 1.89 ms).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "lib")
ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
          "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]


def _newer(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def build_all(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    csrc = os.path.join(HERE, "csrc")
    hdrs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    hdrs += [os.path.join(ROOT, "include", "sdm_c.h"),
             os.path.join(ROOT, "include", "sdm", "ProbabilityMapping.h")]
    hdrs = [h for h in hdrs if os.path.exists(h)]
    outs = []
    eng = os.path.join(csrc, "sdm_engine.hip")
    out = os.path.join(LIB, "libsdm_hip.so")
    if force or _newer(out, [eng] + hdrs):
        cmd = ["hipcc", "--offload-arch=" + ARCH] + COMMON + [eng, "-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    outs.append(out)
    pm = os.path.join(HERE, "host", "ProbabilityMapping.cc")
    if os.path.exists(pm):
        out2 = os.path.join(LIB, "libsdm_pm.so")
        if force or _newer(out2, [pm] + hdrs):
            cmd = ["hipcc"] + COMMON + [pm, "-o", out2, "-L" + LIB, "-lsdm_hip", "-Wl,-rpath,$ORIGIN"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        outs.append(out2)
    return outs


if __name__ == "__main__":
    print("\n".join(build_all(force="--force" in sys.argv, verbose=True)))
