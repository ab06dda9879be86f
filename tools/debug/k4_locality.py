"""K4 with perfect / normal locality: the same instruction stream with every neighbour replaced by (a) the normal covisible
neighbours, (b) one fixed keyframe (all reads hit one map), (c) the reference keyframe itself."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, sdm_pkg, bench
pkg = sdm_pkg.load()
wl = bench.Workload(pkg, torch, "480p", 64, 20, 2.6, 1, 0, 0)
eng, pl = wl.eng, wl.pl
eng.recon(pl["own_slots"], pl["nbr_slots"], wl.min_d, wl.max_d)
def t(nbrs, label):
    for _ in range(5): eng.inter_check_pointset(pl["own_slots"], nbrs)
    eng.enable_timing(True); eng.get_timing(reset=True)
    for _ in range(30): eng.inter_check_pointset(pl["own_slots"], nbrs)
    eng.synchronize(); ms, n = eng.get_timing()["inter"]; eng.enable_timing(False)
    print("%-40s K4+K5 %.4f ms" % (label, ms / n), flush=True)
t(pl["nbr_slots"], "covisible neighbours (bench)")
t([[32] * 20 for _ in pl["own_slots"]], "every neighbour = keyframe 32")
t([[k] * 20 for k in pl["own_slots"]], "every neighbour = the reference itself")
t([[ (k + 1) % 64 ] * 20 for k in pl["own_slots"]], "every neighbour = keyframe k+1")
t(pl["nbr_slots"], "covisible neighbours again")
