"""Can a memory-bound writer (the ingest pre-pass moves ~0.85 GB per 64-keyframe block, nearly all stores) run BESIDE K1 when it
is confined to a few CUs?  A fill of 845 MB (hipMemsetAsync) on a stream created with hipExtStreamCreateWithCUMask against the
engine's step on its own stream: each alone, then together, for several masks.  usage: python tools/debug/cumask_probe.py"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sdm_pkg  # noqa: E402
import bench  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
pkg = sdm_pkg.load()
wl = bench.Workload(pkg, torch, "480p", 64, 20, 2.6, 1, 0, 0)
eng, pl = wl.eng, wl.pl
NBYTES = 845 * 1000 * 1000
buf = torch.empty(NBYTES, dtype=torch.uint8, device="cuda")
ptr = ctypes.c_void_p(buf.data_ptr())


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return s


def fill(s):
    rc = hip.hipMemsetAsync(ptr, 0, ctypes.c_size_t(NBYTES), s)
    assert rc == 0, rc


def sync(s):
    assert hip.hipStreamSynchronize(s) == 0


def step():
    pkg.shard.pipeline_step(eng, None, pl, wl.min_d, wl.max_d, "none", None, "torch")


def k1():
    eng.search_fuse(pl["own_slots"], pl["nbr_slots"], wl.min_d, wl.max_d)


def wall(fn, reps, *streams):
    eng.synchronize()
    for s in streams:
        sync(s)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.synchronize()
    for s in streams:
        sync(s)
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(5):
    step()
print("step alone %.3f ms, K1 alone %.3f ms" % (wall(step, 20), wall(k1, 20)), flush=True)
ALL = [0xFFFFFFFF] * 8
masks = {
    "all 256": ALL,
    "low 32 bits": [0xFFFFFFFF, 0, 0, 0, 0, 0, 0, 0],
    "low 64 bits": [0xFFFFFFFF, 0xFFFFFFFF, 0, 0, 0, 0, 0, 0],
    "every 8th bit (32)": [0x01010101] * 8,
    "every 4th bit (64)": [0x11111111] * 8,
    "every 2nd bit (128)": [0x55555555] * 8,
    "every 16th bit (16)": [0x00010001] * 8,
}
for name, words in masks.items():
    s = masked_stream(words)
    for _ in range(3):
        fill(s)
    sync(s)
    t_fill = wall(lambda: fill(s), 10, s)
    both_step = wall(lambda: (fill(s), step()), 20, s)
    both_k1 = wall(lambda: (fill(s), k1()), 20, s)
    print("%-22s fill alone %.3f ms (%.2f TB/s) | fill + step together %.3f ms | fill + K1 together %.3f ms" % (
        name, t_fill, NBYTES / t_fill / 1e9, both_step, both_k1), flush=True)
    hip.hipStreamDestroy(s)
