"""Runs every device arithmetic self-test (include/sdm_c.h: sdm_selftest) and prints (mismatches, auxiliary count)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdm_pkg  # noqa: E402

pkg = sdm_pkg.load()
eng = pkg.Engine(64, 48, 2)
for which in range(10):
    bad, aux = eng.selftest(which)
    print("selftest %d: mismatches %d, aux %d" % (which, bad, aux))
eng.close()
