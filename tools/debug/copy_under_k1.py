"""Does an H2D copy on a second stream make progress while K1 fills the GPU?  Times a 4.9 MB pinned->device copy (torch, its own
stream; also a high-priority stream) issued right after a K1 launch against the same copy on an idle GPU."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sdm_pkg  # noqa: E402
import bench  # noqa: E402

pkg = sdm_pkg.load()
wl = bench.Workload(pkg, torch, "480p", 64, 20, 2.6, 1, 0, 0)
eng, pl = wl.eng, wl.pl
src = torch.empty(16 * 640 * 480, dtype=torch.uint8).pin_memory()
dst = torch.empty_like(src, device="cuda")
lo, hi = -1, 0
for name, st in (("default-priority stream", torch.cuda.Stream()), ("high-priority stream", torch.cuda.Stream(priority=-1))):
    for busy in (False, True):
        ts = []
        for rep in range(20):
            torch.cuda.synchronize()
            eng.synchronize()
            if busy:
                eng.search_fuse(pl["own_slots"], pl["nbr_slots"], wl.min_d, wl.max_d)  # ~1 ms of K1 on the engine's stream
            t0 = time.perf_counter()
            with torch.cuda.stream(st):
                dst.copy_(src, non_blocking=True)
            st.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            eng.synchronize()
        ts.sort()
        print("%s, GPU %s: copy of %.1f MB done after median %.3f ms (min %.3f)" % (name, "running K1" if busy else "idle", src.numel() / 1e6, ts[len(ts) // 2], ts[0]), flush=True)
