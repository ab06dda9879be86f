#!/usr/bin/env python3
"""bench.py -- throughput of the ProbabilityMapping hot path on MI355X.

One "step" = one pass of the whole path over this rank's block of keyframes:
  SemiDenseRecon (epipolar search + fusion + intra-keyframe check/grow, PM.cc:137-256)
  -> [N>1: RCCL all-gather of the per-keyframe {rho,sigma} maps]
  -> InterKeyFrameDepthChecking (PM.cc:628-799) -> UpdateSemiDensePointSet (PM.cc:337-367)
with every input already resident in HBM (search records packed before the timed region).

Workload at N=1: BASELINE.json configs[1] -- 640x480, 64 keyframes x 20 covisible neighbours,
synthetic gradient images.  Weak scaling: every rank owns --kfs keyframes of one N*kfs sequence.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel k_search_fuse, HIP events on the
engine's stream, algorithmic bytes P*(17+9N) per keyframe) and `cpu_baseline` (the CPU oracle
timed on this box's host cores on a bounded sample; a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kfs", type=int, default=64, help="keyframes per GPU")
    ap.add_argument("--nbrs", type=int, default=20, help="covisible neighbours per keyframe")
    ap.add_argument("--res", default="480p", choices=["480p", "720p", "1080p"])
    ap.add_argument("--disparity", type=float, default=2.6, help="adjacent-keyframe disparity (px): scan-length knob")
    ap.add_argument("--cpu-kfs", type=int, default=48, help="keyframes in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-stats", action="store_true")
    ap.add_argument("--exchange", default="halo", choices=["halo", "allgather"],
                    help="N>1 exchange of {rho,sigma} maps between K3 and K4 (shard.py)")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    import sdm_pkg

    pkg = sdm_pkg.load()
    synth, shard = pkg.synth, pkg.shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    # SDM_BENCH_REHEARSE=1: every rank on GPU 0 with a gloo group and a host-staged exchange -- a dry run of the
    # multi-rank control flow on a one-GPU box; its numbers are not measurements and the JSON says so
    rehearse = os.environ.get("SDM_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    cam = {"480p": synth.TUM1, "720p": synth.HD720, "1080p": synth.HD1080}[args.res]
    W, H, N = cam["W"], cam["H"], args.nbrs
    P = W * H
    n_total = args.kfs * world
    seed = {"480p": 0x5EED0002, "720p": 0x5EED0003, "1080p": 0x5EED0004}[args.res]
    scene = synth.Scene(cam, seed, disparity_px=args.disparity)
    pl = shard.plan(n_total, world, rank, N, scene.neighbours)
    own, nbrs = pl["own"], pl["nbrs"]
    min_d, max_d = scene.depth_prior()

    # ---- engine on torch's current stream; depth pool owned by torch so RCCL gathers it in place
    stream = torch.cuda.Stream()  # a real (non-null) stream shared by torch/RCCL and the engine
    torch.cuda.set_stream(stream)
    pool = torch.zeros((n_total, H, W, 2), dtype=torch.float32, device="cuda")
    eng = pkg.Engine(W, H, n_total, max_neighbours=N, device=local_rank, batch_capacity=min(args.kfs, 64),
                     with_pointset=True, ext_depth_pool=pool.data_ptr(), stream=stream.cuda_stream)
    arch = eng.arch()
    K = scene.K()
    images = {}
    t0 = time.time()
    for k in pl["inputs"]:  # own block + input halo, rendered on the GPU, packed into search records
        im, _ = scene.render(k, device="cuda")
        torch.cuda.synchronize()
        eng.upload_image_device(k, im.data_ptr(), K, scene.Tcw(k))
        if rank == 0 and k < args.cpu_kfs + 2 * N:
            images[k] = im.cpu().numpy()
    t_gen = time.time() - t0
    # PCIe-inclusive variant (reported in DESIGN.md, never `value`): the same keyframes handed over as
    # HOST gray images through sdm_upload_image (H2D copy + device pre-pass + record packing)
    t_h2d = None
    if rank == 0 and world == 1 and images:
        ks = sorted(images)[:16]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in ks:
            eng.upload_image(k, images[k], K, scene.Tcw(k))
        eng.synchronize()
        t_h2d = (time.perf_counter() - t0) / len(ks)

    def step():
        shard.pipeline_step(eng, pool, pl, min_d, max_d, args.exchange)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- scan statistics (untimed counting variant of K1) ------------------------------------------
    stats = None
    if not args.no_stats:
        eng.enable_stats(True)
        eng.get_stats(reset=True)
        eng.search_fuse(own, nbrs, min_d, max_d)
        stats = eng.get_stats()
        eng.enable_stats(False)

    for _ in range(args.warmup):
        step()
    eng.enable_timing(True)
    eng.get_timing(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    timing = eng.get_timing(reset=True)
    eng.enable_timing(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank != 0:
        eng.close()
        if world > 1:
            dist.destroy_process_group()
        return

    ms_step = dt / args.steps * 1e3
    value = P * n_total * args.steps / dt / 1e6
    k1_ms, k1_n = timing["search_fuse"]
    # algorithmic bytes of k_search_fuse (P*(17+9N) per reference keyframe) over the
    # keyframes one launch covers; a step may split its keyframes over 2 launches (boundary/interior)
    k1_avg_ms = k1_ms / max(k1_n, 1)
    k1_bytes = P * (17 + 9 * N) * len(own) * args.steps / max(k1_n, 1)
    achieved = k1_bytes / (k1_avg_ms * 1e-3) / 1e9
    out = {
        "metric": "Mpix*KF/s fused (%dx%dxN_KF)" % (W, H),
        "value": round(value, 2),
        "unit": "Mpix*KF/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "%dx%d, %d keyframes/GPU x %d covisible neighbours, synthetic gradient images "
                        "(BASELINE.json configs[1])" % (W, H, args.kfs, N),
            "stages": "SemiDenseRecon(K1-K3)+%s+InterKFCheck(K4)+PointSet(K5, back-projected inside K4's kernel)" %
                      ("no exchange (1 GPU)" if world == 1 else args.exchange + " exchange of {rho,sigma} maps (RCCL)"),
            "keyframes_total": n_total, "neighbours": N, "disparity_px": args.disparity,
            "parallelism": "keyframe-block x%d" % world, "arch": arch,
        },
        "stage_ms_per_step": {s: round(v[0] / args.steps, 4) for s, v in timing.items()},
        "roofline": {
            "bound": "hbm", "kernel": "k_search_fuse",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": k1_traffic(args, len(own), k1_n, args.steps),
            "launch_ms": round(k1_avg_ms, 4), "launches": k1_n,
            "algorithmic_bytes_per_launch": k1_bytes,
        },
    }
    if rehearse:
        out["rehearsal"] = "all ranks on GPU 0, gloo, host-staged exchange: control-flow dry run, not a measurement"
    if stats:
        out["scan"] = {
            "searches": stats["searches"], "mean_candidates_per_search": round(stats["candidates"] / max(stats["searches"], 1), 3),
            "gate_pass": stats["gate_pass"], "hypotheses": stats["hypotheses"], "fused_pixels": stats["fused"],
            "Mhyp_per_s": round(stats["searches"] / (k1_avg_ms * 1e-3) / 1e6, 1),
        }

    # ---- CPU baseline: the oracle (a port; the reference itself cannot be built) on a bounded sample
    if args.cpu_kfs > 0 and world == 1:  # rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(args, pkg, eng, scene, images, n_total, N, min_d, max_d, W, H)
    out["gen_s"] = round(t_gen, 2)
    if t_h2d is not None:
        out["host_upload_ms_per_keyframe"] = round(t_h2d * 1e3, 4)
        out["value_pcie_inclusive"] = round(P * n_total / (dt / args.steps + t_h2d * len(own)) / 1e6, 2)
    print(json.dumps(out), flush=True)  # the result line first; teardown cannot lose it
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def k1_traffic(args, n_own, k1_launches, steps):
    """HBM-side bytes of one k_search_fuse launch, from the committed PMC run of the same workload
    (profiles/r01_traffic.json, produced by tools/pmc.sh: rocprofv3 cannot run inside bench.py).  null
    when this run's workload differs from the profiled one."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    w = t.get("workload", {})
    if (w.get("res"), w.get("kfs"), w.get("nbrs"), w.get("disparity")) != (args.res, args.kfs, args.nbrs, args.disparity):
        return None
    if k1_launches != steps:  # profiled with one launch per step (single GPU)
        return None
    return t["traffic_bytes_per_launch"]


def cpu_baseline(args, pkg, eng, scene, images, n_total, N, min_d, max_d, W, H):
    """Times oracle/pm_oracle.c (checker + CPU baseline ONLY; never on the product path) on the
    first cpu_kfs keyframes of the same workload, same stages as the GPU step: (i) 1 thread = the
    reference's effective behaviour (its OpenMP pragmas are inert), (ii) OpenMP on
    all host cores (what PM.cc:197's pragma intends)."""
    import ctypes
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pm_oracle
    out_dir = os.path.join(ROOT, "gpurun_out", "_oracle_build")
    ref = pm_oracle.Oracle("omp", out_dir=out_dir)  # -O3 -march=native -fopenmp, built on THIS box
    ncores = ref.num_threads()
    sample = [k for k in range(args.cpu_kfs) if all(j in images for j in scene.neighbours(k, n_total, N))]
    need = sorted(set(sample) | {j for k in sample for j in scene.neighbours(k, n_total, N)})
    kf, gpu_maps = {}, {}
    for k in need:
        g, t, s = ref.gradient_prepass(images[k])
        kf[k] = ref.keyframe(images[k], g, t, s, scene.K(), scene.Tcw(k))
        gpu_maps[k] = eng.download_depth(k)  # neighbours' finished maps for the inter-keyframe check

    def run():
        maps, chk = {}, {}
        st_tot = dict(searches=0, candidates=0)
        for k in sample:
            nb = scene.neighbours(k, n_total, N)
            r, s, st = ref.semi_dense_recon(kf[k], [kf[j] for j in nb], None, min_d, max_d)
            maps[k] = (r, s)
            st_tot["searches"] += st["searches"]
            st_tot["candidates"] += st["candidates"]
        for k in sample:
            nb = scene.neighbours(k, n_total, N)
            chk[k] = ref.inter_check(kf[k], maps[k][0], [kf[j] for j in nb], [gpu_maps[j][0] for j in nb],
                                     [gpu_maps[j][1] for j in nb])
            ref.pointset(kf[k], chk[k])
        return maps, chk, st_tot

    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(1)
    except OSError:
        omp = None
    t0 = time.perf_counter()
    maps, chk, st = run()
    t1 = time.perf_counter() - t0
    tn = t1
    if omp is not None:
        omp.omp_set_num_threads(ncores)
        t0 = time.perf_counter()
        run()
        tn = time.perf_counter() - t0
    # live parity check of the GPU result against the oracle on the sample (depth L1 vs ref)
    l1, nmask, mism = 0.0, 0, 0
    for k in sample:
        for got, want in ((gpu_maps[k][0], maps[k][0]), (eng.download_checked(k), chk[k])):
            m = (want > 1e-6)
            mism += int(((got > 1e-6) != m).sum())
            l1 += float(np.abs(got[m] - want[m]).sum())
            nmask += int(m.sum())
    px = W * H * len(sample)
    return {
        "value": round(px / t1 / 1e6, 3), "unit": "Mpix*KF/s", "cores": 1, "kind": "port",
        "sample": "same stages (K1-K5) on the first %d keyframes of the same workload, oracle/pm_oracle.c "
                  "-O3 -march=native, 1 thread (the reference's OpenMP pragmas are inert)" % len(sample),
        "seconds": round(t1, 2),
        "all_cores": {"value": round(px / tn / 1e6, 3), "cores": ncores if omp is not None else 1,
                      "seconds": round(tn, 2)},
        "mean_candidates_per_search": round(st["candidates"] / max(st["searches"], 1), 3),
        "parity_on_sample": {"mask_mismatches": mism, "depth_L1": (l1 / max(nmask, 1)), "pixels": nmask},
    }


if __name__ == "__main__":
    main()
