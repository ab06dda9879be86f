#!/usr/bin/env python3
"""bench.py -- throughput of the ProbabilityMapping hot path on MI355X.

One "step" = one pass of the whole path over this rank's block of keyframes:
  SemiDenseRecon (epipolar search + fusion + intra-keyframe check/grow, PM.cc:137-256)
  -> [N>1: exchange of the per-keyframe {rho,sigma} maps over RCCL: all-gather of the maps that cross ranks (default),
      all-gather of the whole block, or point-to-point halo -- all three are timed, `value` is the --exchange one]
  -> InterKeyFrameDepthChecking (PM.cc:628-799) -> UpdateSemiDensePointSet (PM.cc:337-367)
with every input already resident in HBM (search records packed before the timed region).

`value` is measured on BASELINE.json configs[1] -- 640x480, 64 keyframes x 20 covisible neighbours, synthetic gradient
images -- in steady state (after a 0.3 s pre-warm phase; the same W + K steps straight after set-up are reported as
value_cold / ms_per_step_cold / roofline.frac_cold).  At N=1 the same JSON line carries `extra_configs`: the
north_star's target case (640x480 x 256 keyframes, N=20), configs[2] (1280x720 x 256 keyframes, N=7), configs[1] with two
outlier neighbours per keyframe (K1 only), configs[1] on i.i.d.-noise images (SURVEY.md §8d adversarial set) and
configs[1] with a long baseline (10 px/keyframe), each with its own ms_per_step, K1 roofline and mean scan length.
Weak scaling: every rank owns --kfs keyframes of one N*kfs sequence (--independent: one separate sequence per GPU,
configs[4]).

`python bench.py --gpus N` with N>1 starts its own `torch.distributed.run` child (before anything touches the GPU) when it
was not launched by one; the line printed is the child's.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel k_search_fuse + its follow-up k_fuse_open, HIP events on
the engine's stream, algorithmic bytes P*(17+9N) per keyframe; `bound` = what the counters say limits it),
`step_roofline` (all kernels' algorithmic bytes over the step time) and `cpu_baseline` (the CPU oracle timed on this
box's host cores on a bounded sample; a reported baseline, not the target).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
SEEDS = {"480p": 0x5EED0002, "720p": 0x5EED0003, "1080p": 0x5EED0004}
BASELINE_CONFIG = {("480p", 64, 20): "BASELINE.json configs[1]",
                   ("480p", 256, 20): "north_star target case",
                   ("720p", 256, 7): "BASELINE.json configs[2]",
                   ("480p", 8, 7): "BASELINE.json configs[0] geometry"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kfs", type=int, default=64, help="keyframes per GPU")
    ap.add_argument("--nbrs", type=int, default=20, help="covisible neighbours per keyframe")
    ap.add_argument("--res", default="480p", choices=["480p", "720p", "1080p"])
    ap.add_argument("--disparity", type=float, default=2.6, help="adjacent-keyframe disparity (px): scan-length knob")
    ap.add_argument("--prior-spread", type=float, default=0.1,
                    help="depth prior of StereoSearchConstraints (PM.cc:381-382): s = spread * mu (App. D: 0.1; ORB depths of "
                         "a real keyframe spread 0.3-0.5) -- the other scan-length knob")
    ap.add_argument("--scene", default="plane", choices=["plane", "strip"],
                    help="strip: App. D's second plane -- a foreground strip with occluding edges (depth discontinuities)")
    ap.add_argument("--roll", type=float, default=1.0,
                    help="keyframes' in-plane roll drawn from [-roll, roll] degrees (App. D: 1); above 1 the pairs carry the "
                         "matching median in-plane rotation of PM.cc:170-179")
    ap.add_argument("--cpu-kfs", type=int, default=48, help="keyframes in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-stats", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs runs (N=1 only)")
    ap.add_argument("--no-streaming", action="store_true", help="skip the sustained ingest + compute figure (N=1 only)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 child runs (then: the committed PMC file, or null)")
    ap.add_argument("--exchange", default="allgather", choices=["halo", "allgather", "allgather_late", "allgather_full"],
                    help="N>1: the exchange of {rho,sigma} maps between K3 and K4 that `value` is measured with: "
                         "allgather = RCCL all-gather of the maps that cross ranks (every rank's boundary keyframes, "
                         "overlapped with the reconstruction of the interior ones); allgather_full = every rank's whole "
                         "block, pipelined in sub-blocks (BASELINE.json's literal wording, 3x the bytes); allgather_late = the boundary "
                         "all-gather after an UNSPLIT reconstruction, hidden behind the local keyframes' K4 only; halo = "
                         "point-to-point.  The other forms are timed too: exchange_ms_per_step, value_<form>")
    ap.add_argument("--wire", default="compact", choices=["compact", "whole"],
                    help="native transport: what crosses ranks per depth map -- the {rho,sigma} of the keyframe's active-list "
                         "entries (the map is zero elsewhere; the receiver holds the same list), or the whole map")
    ap.add_argument("--transport", default="native", choices=["native", "torch"],
                    help="N>1: RCCL called by the engine's C ABI (sdm_exchange_*) or torch.distributed on the pool tensor")
    ap.add_argument("--noise", action="store_true", help="i.i.d. uniform u8 images (SURVEY.md §8d adversarial set)")
    ap.add_argument("--outliers", type=int, default=0,
                    help="N=1: this many of every keyframe's neighbours are wrong-pose copies (K1 only is stepped: the "
                         "profiling form of the extra_configs outlier line)")
    ap.add_argument("--independent", action="store_true",
                    help="N>1: one independent sequence per GPU, no exchange (BASELINE.json configs[4])")
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 from a bare shell: become the parent of `python -m torch.distributed.run ... bench.py`.
    Nothing in this process has touched the GPU (torch is not even imported yet); the child is a child
    process, never an exec."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env, cwd=ROOT)


def under_profiler():
    return any(("rocprof" in os.environ.get(k, "").lower()) for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")) or \
        any(k.startswith("ROCPROF") for k in os.environ)


LIVE_PMC_GROUPS = (("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"),
                   ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"))


def live_traffic(args):
    """HBM-side bytes of one k_search_fuse launch, measured in THIS run: one `rocprofv3 --pmc` child per counter group (no
    trace domains; the read-request sizes and the write requests cannot share a pass) over two steps of the same workload,
    started before this process touches the GPU (a GPU process must not exec, and two processes would share the card).
    bytes = 32 n32 + 64 n64 + 128 n128 (+ 64 w64 + 32 (w - w64)): TCC_EA0 counters summed over channels, averaged over the
    kernel's dispatches -- tools/pmc_summary.py's arithmetic, the guide's "calibrate on your own access pattern".
    Returns (record or None, note)."""
    import csv
    import glob
    import shutil
    import signal
    import tempfile
    if args.no_live_pmc or os.environ.get("SDM_BENCH_PMC_CHILD") or under_profiler():
        return None, "live PMC passes skipped (flag, child run, or already under a profiler)"
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    if not os.path.exists("/dev/kfd"):
        return None, "no GPU device node"
    flags = ["--kfs", str(args.kfs), "--nbrs", str(args.nbrs), "--res", args.res, "--disparity", repr(args.disparity),
             "--prior-spread", repr(args.prior_spread), "--scene", args.scene, "--roll", repr(args.roll),
             "--outliers", str(args.outliers)] + (["--noise"] if args.noise else [])
    child = ["python3", os.path.abspath(__file__)] + flags + ["--steps", "2", "--warmup", "1", "--cpu-kfs", "0", "--no-stats",
                                                               "--no-extra", "--no-streaming", "--no-live-pmc"]
    env = dict(os.environ, TMPDIR="/tmp", SDM_BENCH_PMC_CHILD="1")
    out = tempfile.mkdtemp(prefix="sdm_live_pmc_", dir="/tmp")
    acc = {}
    try:
        for g, group in enumerate(LIVE_PMC_GROUPS):
            d = os.path.join(out, "p%d" % g)
            cmd = [exe, "--pmc"] + list(group) + ["--output-format", "csv", "-d", d, "--"] + child
            with open(os.path.join(out, "p%d.log" % g), "wb") as log:
                pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, start_new_session=True)
                try:
                    rc = pr.wait(timeout=240)
                except subprocess.TimeoutExpired:
                    os.killpg(pr.pid, signal.SIGKILL)  # the process group this call started, nothing else
                    pr.wait()
                    return None, "live PMC pass %d timed out" % g
            if rc != 0:
                return None, "live PMC pass %d exited with %d" % (g, rc)
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    if "k_search_fuse<false" in r["Kernel_Name"]:
                        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        c = {k: sum(v) / len(v) for k, v in acc.items()}
        need = LIVE_PMC_GROUPS[0] + LIVE_PMC_GROUPS[1]
        if not all(k in c for k in need):
            return None, "live PMC passes returned no k_search_fuse rows for %s" % [k for k in need if k not in c]
        rd = 32 * c[need[0]] + 64 * c[need[1]] + 128 * c[need[2]]
        wr = 64 * c["TCC_EA0_WRREQ_64B_sum"] + 32 * (c["TCC_EA0_WRREQ_sum"] - c["TCC_EA0_WRREQ_64B_sum"])
        return {"traffic": rd + wr, "read": rd, "write": wr, "dispatches": len(acc[need[0]])}, \
            "live: %d rocprofv3 --pmc passes (TCC_EA0 read-request sizes; write requests) over a 2-step child run of this " \
            "workload, started by this bench.py before it touched the GPU; averaged over %d dispatches" % (
                len(LIVE_PMC_GROUPS), len(acc[need[0]]))
    except Exception as e:  # never let the measurement of one field break the bench line
        return None, "live PMC passes failed: %r" % (e,)
    finally:
        shutil.rmtree(out, ignore_errors=True)


def source_hash():
    """hash of the kernel sources (device code: sdm_device.h, sdm_kernels.h, sdm_ingest.h): ties a committed PMC traffic figure to the
    kernels it was measured on; the host side of the engine and the exchange (sdm_engine.hip, sdm_comm.h) are not part of it"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "orb-slam-free-space-carving_amd", "csrc")
    for f in ("sdm_device.h", "sdm_kernels.h", "sdm_ingest.h"):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def workload_name(W, H, kfs, N, res, independent=False):
    tag = BASELINE_CONFIG.get((res, kfs, N))
    s = "%dx%d, %d keyframes/GPU x %d covisible neighbours, synthetic gradient images" % (W, H, kfs, N)
    if independent:
        s += ", one independent sequence per GPU (BASELINE.json configs[4])"
    return s + (" (%s)" % tag if tag and not independent else "")


class Workload:
    """One resident workload on this rank: scene, plan, engine, uploaded keyframes."""

    def __init__(self, pkg, torch, res, kfs, N, disparity, world, rank, local_rank, independent=False,
                 keep_images=0, noise=False, outliers=0, spread=0.1, strip=False, roll=1.0):
        synth, shard = pkg.synth, pkg.shard
        self.pkg, self.torch = pkg, torch
        cam = {"480p": synth.TUM1, "720p": synth.HD720, "1080p": synth.HD1080}[res]
        self.res, self.W, self.H, self.N, self.kfs = res, cam["W"], cam["H"], N, kfs
        self.P = self.W * self.H
        self.independent = independent
        if independent:  # configs[4]: every GPU has its own sequence (seeds 0x5EED0050..57), no exchange
            self.n_total = kfs
            self.scene = synth.Scene(cam, 0x5EED0050 + rank, disparity_px=disparity, noise_images=noise, strip=strip,
                                     roll_deg=roll)
            self.pl = shard.plan(kfs, 1, 0, N, self.scene.neighbours)
        else:
            self.n_total = kfs * world
            self.scene = synth.Scene(cam, SEEDS[res], disparity_px=disparity, noise_images=noise, strip=strip, roll_deg=roll)
            self.pl = shard.plan(self.n_total, world, rank, N, self.scene.neighbours)
        pl = self.pl
        self.strip, self.roll = strip, roll
        self.rots = None
        if roll != 1.0:  # the pairs' median in-plane rotations (PM.cc:170-179), per own keyframe and neighbour
            self.rots = [[self.scene.rot_deg(k, j) for j in pl["nbrs"][i]] for i, k in enumerate(pl["own"])]
            self.pl = pl = dict(pl, rot_of=dict(zip(pl["own_slots"], self.rots)))
        self.outliers = outliers
        n_slots = pl["n_slots"] * (2 if outliers else 1)
        self.spread = spread
        self.min_d, self.max_d = self.scene.depth_prior(spread)
        # engine on torch's current stream; depth pool owned by torch (the torch transport exchanges it in place)
        self.pool = torch.zeros((n_slots, self.H, self.W, 2), dtype=torch.float32, device="cuda")
        self.eng = pkg.Engine(self.W, self.H, n_slots, max_neighbours=N, device=local_rank,
                              batch_capacity=min(kfs, 64), with_pointset=True, ext_depth_pool=self.pool.data_ptr(),
                              stream=torch.cuda.current_stream().cuda_stream)
        self.K = self.scene.K()
        self.images = {}
        t0 = time.time()
        for k in pl["inputs"]:  # own block + input halo, rendered on the GPU, packed into search records
            im, _ = self.scene.render(k, device="cuda")
            torch.cuda.synchronize()
            self.eng.upload_image_device(pl["slot"][k], im.data_ptr(), self.K, self.scene.Tcw(k))
            if k < keep_images:
                self.images[k] = im.cpu().numpy()
            if outliers:  # a wrong-pose copy of every keyframe (baseline stretched by 40 %) in the upper half of the slots
                T = self.scene.Tcw(k).copy()
                T[:, 3] *= 1.4
                self.eng.upload_image_device(pl["n_slots"] + pl["slot"][k], im.data_ptr(), self.K, T)
        if outliers:
            # `outliers` of every keyframe's neighbours (list positions 3, 7, 11, ...) are redirected to the wrong-pose
            # copy: their hypotheses are consistent-looking outliers, so the first accepted hypothesis is not compatible
            # with all others and the fusion's shortcut does not settle the pixel
            nb = [list(r) for r in pl["nbr_slots"]]
            for r in nb:
                for i in range(outliers):
                    pos = min(3 + 4 * i, len(r) - 1)
                    r[pos] = pl["n_slots"] + r[pos]
            self.pl = dict(pl, nbr_slots=nb)
        self.t_gen = time.time() - t0

    def step(self, exchange, transport, group=None):
        if self.outliers:  # K1 only: the wrong-pose copies are neighbours, never reference keyframes (no maps for K4)
            self.eng.search_fuse(self.pl["own_slots"], self.pl["nbr_slots"], self.min_d, self.max_d, rot=self.rots)
            return
        self.pkg.shard.pipeline_step(self.eng, self.pool, self.pl, self.min_d, self.max_d, exchange, group, transport)

    def scan_stats(self):
        eng, pl = self.eng, self.pl
        eng.enable_stats(True)
        eng.get_stats(reset=True)
        eng.search_fuse(pl["own_slots"], pl["nbr_slots"], self.min_d, self.max_d, rot=self.rots)
        st = eng.get_stats()
        eng.enable_stats(False)
        return st

    def close(self):
        self.eng.close()
        self.pool = None
        self.torch.cuda.empty_cache()


PREWARM_S = 0.3  # the GPU's clocks and caches need ~0.2 s of work to settle (tools/k1_time.py: the first 20 launches
                 # after a cold start run 8 % slower than every later round); this setup phase is neither warm-up nor timed


def prewarm(wl, barrier, exchange, transport, reduce_max=None):
    """steady-state clocks before the contract's W warm-up steps: untimed steps for about PREWARM_S seconds.  The
    step count is derived from the (max over ranks) time of the first four steps, so that every rank runs the same
    number of collective steps."""
    barrier()
    t0 = time.perf_counter()
    for _ in range(4):
        wl.step(exchange, transport)
    wl.torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 4
    if reduce_max is not None:
        t = reduce_max(t)
    n = int(min(400, max(4, PREWARM_S / max(t, 1e-5))))
    if os.environ.get("SDM_BENCH_PMC_CHILD"):
        n = 4  # counter passes: every dispatch is serialised and read out, and clocks do not change byte counts
    for _ in range(n):
        wl.step(exchange, transport)
    wl.torch.cuda.synchronize()
    return n + 4


def timed(wl, steps, warmup, barrier, exchange, transport):
    """W warm-up steps, then exactly `steps` steps between barrier + synchronize on both sides."""
    for _ in range(warmup):
        wl.step(exchange, transport)
    wl.eng.enable_timing(True)
    wl.eng.get_timing(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step(exchange, transport)
    barrier()
    dt = time.perf_counter() - t0
    timing = wl.eng.get_timing(reset=True)
    wl.eng.enable_timing(False)
    return dt, timing


def roofline(wl, timing, steps, traffic, traffic_source=None):
    """K1's roofline record.  `achieved` = ALGORITHMIC bytes (P*(17+9N) per reference
    keyframe x the keyframes one launch covers) / the launch's HIP-event duration; `traffic` = HBM-side bytes
    per launch from the PMC run of the same build and workload (null otherwise); hbm_GBs = traffic / duration."""
    k1_ms, k1_n = timing["search_fuse"]
    k1_avg_ms = k1_ms / max(k1_n, 1)
    k1_bytes = wl.P * (17 + 9 * wl.N) * len(wl.pl["own"]) * steps / max(k1_n, 1)
    achieved = k1_bytes / (k1_avg_ms * 1e-3) / 1e9
    out = {
        # what the counters say bounds the kernel (profiles/*_pmc.txt): the vector ALU's issue slots are
        # ~85 % full while the HBM pins carry ~0.2 of their peak.  achieved / peak / frac stay SURVEY.md §8d's figure:
        # ALGORITHMIC bytes per launch over the launch time, against the HBM peak.
        # `bound` names the roofline `achieved` / `peak` / `frac` are priced against (the bench contract's "hbm" | "mfma");
        # `limiter` = what the kernel's own counters say holds it back
        "bound": "hbm", "model": "hbm (SURVEY.md §8d algorithmic bytes / HBM peak)", "kernel": "k_search_fuse",
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "launch_ms": round(k1_avg_ms, 4), "launches": k1_n,
        "algorithmic_bytes_per_launch": k1_bytes,
        "limiter": "valu-issue",
    }
    if traffic:
        out["hbm_GBs"] = round(traffic / (k1_avg_ms * 1e-3) / 1e9, 1)
        out["hbm_frac"] = round(out["hbm_GBs"] / HBM_PEAK_GBS, 4)
        out["traffic_source"] = traffic_source or \
            "profiles/ PMC run (rocprofv3 --pmc, TCC_EA0 request counters) of this build and workload"
    return out, k1_avg_ms


def step_roofline(wl, ms_step):
    """all kernels of the step: SURVEY.md §8d's algorithmic bytes K1 P(17+9N) + K2 16P + K3 20P + K4 P(8+8N) + K5 16P =
    P(77+17N) per keyframe, over the measured step time, against the HBM peak"""
    b = wl.P * (77 + 17 * wl.N) * len(wl.pl["own"])
    gbs = b / (ms_step * 1e-3) / 1e9
    return {"algorithmic_bytes_per_step": b, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4),
            "note": "the dense-plane byte model overstates the list kernels' work: K2-K5 touch only the ~20 % of the "
                    "pixels the reference touches (PM.cc:201, 662); their own counters are in profiles/"}


def measure(wl, steps, warmup, barrier, exchange, transport, reduce_max=None):
    """the contract's W warm-up + K timed steps twice: straight after set-up (cold clocks: *_cold) and again after the
    pre-warm phase (steady state: the figures `value` / `roofline` report)"""
    dt_c, timing_c = timed(wl, steps, warmup, barrier, exchange, transport)
    n_pre = prewarm(wl, barrier, exchange, transport, reduce_max)
    dt, timing = timed(wl, steps, warmup, barrier, exchange, transport)
    if reduce_max is not None:
        dt_c, dt = reduce_max(dt_c), reduce_max(dt)
    return dt, timing, dt_c, timing_c, n_pre


def committed_traffic(res, kfs, nbrs, disparity, k1_launches, steps, noise=False, outliers=0, spread=0.1, strip=False, roll=1.0):
    """HBM-side bytes of one k_search_fuse launch from a committed PMC run (tools/pmc.sh: rocprofv3 cannot
    run inside bench.py).  Only a file whose workload AND kernel-source hash match this build is used."""
    src = source_hash()
    pdir = os.path.join(ROOT, "profiles")
    if k1_launches != steps or not os.path.isdir(pdir):  # profiled with one launch per step (single GPU)
        return None
    for f in sorted(os.listdir(pdir), reverse=True):
        if not (f.endswith(".json") and "traffic" in f):
            continue
        try:
            t = json.load(open(os.path.join(pdir, f)))
        except ValueError:
            continue
        w = t.get("workload", {})
        if t.get("src_hash") == src and \
                (w.get("res"), w.get("kfs"), w.get("nbrs"), w.get("disparity"), bool(w.get("noise", False)),
                 int(w.get("outliers", 0)), float(w.get("spread", 0.1)), w.get("scene", "plane"), float(w.get("roll", 1.0))) == (
                    res, kfs, nbrs, disparity, bool(noise), int(outliers), float(spread), "strip" if strip else "plane", float(roll)):
            return t["traffic_bytes_per_launch"]
    return None


def scan_record(stats, k1_avg_ms):
    return {
        "searches": stats["searches"],
        "mean_candidates_per_search": round(stats["candidates"] / max(stats["searches"], 1), 3),
        "gate_pass": stats["gate_pass"], "hypotheses": stats["hypotheses"], "fused_pixels": stats["fused"],
        "Mhyp_per_s": round(stats["searches"] / (k1_avg_ms * 1e-3) / 1e6, 1),
        # how K1 walked the ranges: share of the (wave, neighbour) scans that took the gradient-mask scan
        "gate_pass_rate": round(stats["gate_pass"] / max(stats["candidates"], 1), 4),
        "mask_scan_waves": stats.get("mask_waves", 0), "mask_scan_steps": stats.get("mask_steps", 0),
    }


def run_extra(pkg, torch, res, kfs, N, disparity, steps, warmup, local_rank, barrier, noise=False, outliers=0, tag=None,
              spread=0.1, strip=False, roll=1.0):
    """one more single-GPU workload, measured the same way as the headline one (cold and steady state)"""
    wl = Workload(pkg, torch, res, kfs, N, disparity, 1, 0, local_rank, noise=noise, outliers=outliers, spread=spread,
                  strip=strip, roll=roll)
    stats = wl.scan_stats()
    dt, timing, dt_c, timing_c, _ = measure(wl, steps, warmup, barrier, "halo", "torch")
    plain = not (noise or outliers or strip) and spread == 0.1 and roll == 1.0
    rf, k1_avg = roofline(wl, timing, steps, committed_traffic(res, kfs, N, disparity, timing["search_fuse"][1], steps,
                                                               noise=noise, outliers=outliers, spread=spread, strip=strip,
                                                               roll=roll))
    rf_c, _ = roofline(wl, timing_c, steps, None)
    rf["frac_cold"] = rf_c["frac"]
    rf["launch_ms_cold"] = rf_c["launch_ms"]
    name = workload_name(wl.W, wl.H, kfs, N, res) if plain and disparity == 2.6 else \
        "%dx%d, %d keyframes x %d neighbours: %s" % (wl.W, wl.H, kfs, N, tag)
    out = {
        "workload": name,
        "value": round(wl.P * wl.n_total * steps / dt / 1e6, 2), "unit": "Mpix*KF/s",
        "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4),
        "ms_per_step_cold": round(dt_c / steps * 1e3, 4),
        "keyframes_total": wl.n_total, "neighbours": N, "disparity_px": disparity, "prior_spread": spread,
        "scene": "strip" if strip else "plane", "roll_deg": roll,
        "stage_ms_per_step": {s: round(v[0] / steps, 4) for s, v in timing.items()},
        "roofline": rf,
        "mean_candidates_per_search": round(stats["candidates"] / max(stats["searches"], 1), 3),
        "scan": scan_record(stats, k1_avg),
    }
    if outliers:
        out["stages"] = "K1 only (k_search_fuse): %d of every keyframe's %d neighbours are wrong-pose copies" % (outliers, N)
        del out["value"], out["unit"]
    else:
        out["step_roofline"] = step_roofline(wl, dt / steps * 1e3)
    wl.close()
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    live, live_note = (None, "multi-rank run") if (args.gpus > 1 or args.independent) else live_traffic(args)

    import torch
    import torch.distributed as dist
    import sdm_pkg

    pkg = sdm_pkg.load()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    # SDM_BENCH_REHEARSE=1: every rank on GPU 0 with a gloo group and a host-staged exchange -- a dry run of the
    # multi-rank control flow on a one-GPU box; its numbers are not measurements and the JSON says so
    rehearse = os.environ.get("SDM_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    transport = args.transport
    rccl_world = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            transport = "torch"  # RCCL refuses two ranks on one device
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
            rccl_world = dist.get_world_size()

    stream = torch.cuda.Stream()  # a real (non-null) stream shared by torch/RCCL and the engine
    torch.cuda.set_stream(stream)

    def barrier():
        if world > 1:
            # drain this rank's queues first: the engine's exchange runs on its own RCCL communicator, and collectives of
            # two communicators must not be in flight together (ranks could start them in different orders)
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    wl = Workload(pkg, torch, args.res, args.kfs, args.nbrs, args.disparity, world, rank, local_rank,
                  independent=args.independent, noise=args.noise, outliers=(args.outliers if world == 1 else 0),
                  spread=args.prior_spread, strip=(args.scene == "strip"), roll=args.roll,
                  keep_images=max(args.cpu_kfs + 2 * args.nbrs, min(args.kfs, 64)) if (rank == 0 and world == 1) else 0)
    eng, pl, W, H, N, P = wl.eng, wl.pl, wl.W, wl.H, wl.N, wl.P
    n_total = wl.n_total * (world if args.independent else 1)
    arch = eng.arch()
    exchanging = world > 1 and not args.independent
    transport_note = None
    wire_entries = 0
    if exchanging and transport == "native":
        # the engine's own RCCL communicator (include/sdm_c.h sdm_comm_init) and one exchange of each form as a
        # self-check.  If any rank fails (library, communicator or transfer error) ALL ranks agree -- over the
        # torch.distributed group -- to fall back to the torch transport, and the JSON line says so.
        ok, why = 1, ""

        def agreed(ok):
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        try:
            pkg.shard.setup_native_comm(eng)
        except Exception as e:  # noqa: BLE001 -- reported, never swallowed
            ok, why = 0, repr(e)
        if agreed(ok):  # every rank holds a communicator: only now may anyone post a transfer
            try:
                if args.wire == "compact":
                    wire_entries = pkg.shard.agree_compact_wire(eng, pl)  # one all-reduce; the same value on every rank
                wl.step("halo", "native")
                wl.step("allgather", "native")
                wl.step("allgather_full", "native")
                torch.cuda.synchronize()
                if wire_entries > 0 and eng.exchange_mismatches():
                    raise RuntimeError("compact wire format: a received keyframe's active list differs from the sender's")
            except Exception as e:  # noqa: BLE001
                ok, why = 0, repr(e)
            ok = 1 if agreed(ok) else 0
        else:
            ok = 0
        if ok == 1:
            rccl_world = eng.comm_info()[0]
        else:
            transport = "torch"
            wire_entries = 0  # the torch transport exchanges whole maps
            transport_note = "native RCCL exchange failed on at least one rank (%s): fell back to torch.distributed" % (
                why or "another rank")
            try:
                eng.comm_destroy()
            except Exception:  # noqa: BLE001
                pass
            eng.compact_entries = 0  # the fall-back moves whole maps (shard.pipeline_step looks at this)
            try:
                eng.exchange_compact(0)
            except Exception:  # noqa: BLE001
                pass

    if exchanging and rehearse and args.wire == "compact":
        # the rehearsal moves the compact payloads as well: packed and scattered by the engine's kernels, staged through
        # host memory, gloo in between (shard.EngineCompactCodec) -- the wire format's control flow, not its speed
        wire_entries = pkg.shard.agree_compact_wire(eng, pl)

    # PCIe-inclusive variant (reported in DESIGN.md, never `value`): the same keyframes handed over as HOST gray images
    # (H2D copy + device pre-pass + record packing + pixel lists): one sdm_upload_images_batch call for the block -- from
    # ordinary (pageable) arrays through the engine's pinned ring, and from pinned memory (sdm_host_alloc) read in place
    # -- and, for comparison, one sdm_upload_image call per keyframe
    upload_ms = None
    if rank == 0 and world == 1 and wl.images:
        ks = [k for k in pl["own"] if k in wl.images][:64]
        slots = [pl["slot"][k] for k in ks]
        ims = [wl.images[k] for k in ks]
        poses = [wl.scene.Tcw(k) for k in ks]
        block = eng.host_alloc((len(ks), H, W))  # a frame queue in pinned memory: one block, frames back to back
        pinned = [block[i] for i in range(len(ks))]
        for a, im in zip(pinned, ims):
            a[...] = im

        def timed_upload(fn, reps=3):
            best = None
            for _ in range(reps + 1):  # the first pass is a warm-up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                eng.synchronize()
                dt_ = (time.perf_counter() - t0) / len(ks)
                best = dt_ if best is None else min(best, dt_)
            return best

        def single():
            for s_, im, T in zip(slots, ims, poses):
                eng.upload_image(s_, im, wl.K, T)

        upload_ms = {
            "batch_pageable": timed_upload(lambda: eng.upload_images_batch(slots, ims, wl.K, poses)) * 1e3,
            "batch_pinned": timed_upload(lambda: eng.upload_images_batch(slots, pinned, wl.K, poses)) * 1e3,
            "per_keyframe_calls": timed_upload(single, reps=1) * 1e3,
            "keyframes": len(ks),
        }
        eng.host_free(block)

    streaming = None
    if rank == 0 and world == 1 and wl.images and not args.outliers and not args.no_streaming:
        import gc
        gc.collect()
        gc.disable()  # (a full collection of this interpreter's heap costs ~40 ms: profiles/r05_upload_spikes.txt)
        try:
            streaming = {"pageable": streaming_rate(pkg, torch, wl), "pinned": streaming_rate(pkg, torch, wl, pinned=True)}
        except Exception as e:  # noqa: BLE001 -- reported in the line, the headline must survive
            streaming = {"error": repr(e)}
        gc.enable()

    stats = None if args.no_stats else wl.scan_stats()  # untimed counting variant of K1

    def reduce_max(dt):
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return dt

    dt, timing, dt_cold, timing_cold, n_prewarm = measure(wl, args.steps, args.warmup, barrier, args.exchange, transport,
                                                          reduce_max if world > 1 else None)
    exchange_ms, others = None, {}
    if exchanging:  # the other exchange forms, same K steps, so the line carries all three
        exchange_ms = {args.exchange: round(dt / args.steps * 1e3, 4)}
        for other in ("allgather", "allgather_late", "allgather_full", "halo"):
            if other == args.exchange:
                continue
            dt2, _ = timed(wl, args.steps, min(args.warmup, 2), barrier, other, transport)
            others[other] = reduce_max(dt2)
            exchange_ms[other] = round(others[other] / args.steps * 1e3, 4)
        if wire_entries > 0:  # and the default form once more with WHOLE maps on the wire, for comparison
            eng.exchange_compact(0)
            dt2, _ = timed(wl, args.steps, min(args.warmup, 2), barrier, args.exchange, transport)
            exchange_ms[args.exchange + ", whole maps on the wire"] = round(reduce_max(dt2) / args.steps * 1e3, 4)
            eng.exchange_compact(wire_entries)

    if rank != 0:
        eng.close()
        if world > 1:
            dist.destroy_process_group()
        return

    ms_step = dt / args.steps * 1e3
    value = P * n_total * args.steps / dt / 1e6
    committed = committed_traffic(args.res, args.kfs, N, args.disparity, timing["search_fuse"][1], args.steps,
                                  noise=args.noise, outliers=args.outliers, spread=args.prior_spread,
                                  strip=(args.scene == "strip"), roll=args.roll) if world == 1 else None
    if live and timing["search_fuse"][1] == args.steps:  # (one K1 launch per step, as in the child run)
        rf, k1_avg_ms = roofline(wl, timing, args.steps, live["traffic"], live_note)
        rf["traffic_read"], rf["traffic_write"] = live["read"], live["write"]
        if committed:
            rf["traffic_committed_profile"] = committed  # the same figure from profiles/*_traffic.json (same source hash)
    else:
        rf, k1_avg_ms = roofline(wl, timing, args.steps, committed)
        rf["traffic_note"] = live_note
    rf_cold, _ = roofline(wl, timing_cold, args.steps, None)
    rf["frac_cold"] = rf_cold["frac"]  # the same K steps after only the W warm-up steps (no pre-warm phase)
    rf["launch_ms_cold"] = rf_cold["launch_ms"]
    if not exchanging:
        xdesc = "no exchange (%s)" % ("1 GPU" if world == 1 else "independent sequences")
    else:
        xdesc = "%s of {rho,sigma} maps over RCCL (%s)" % (
            {"allgather": "all-gather of the boundary keyframes' maps (the ones other ranks read)",
             "allgather_full": "all-gather of every rank's whole block", "halo": "point-to-point halo exchange"}[args.exchange],
            "engine C ABI sdm_allgather_* / sdm_exchange_*" if transport == "native" else "torch.distributed")
    out = {
        "metric": "Mpix*KF/s fused (%dx%dxN_KF)" % (W, H),
        "value": round(value, 2),
        "unit": "Mpix*KF/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4),
        # the same W + K steps run straight after set-up, before the pre-warm phase (clocks / caches not settled)
        "ms_per_step_cold": round(dt_cold / args.steps * 1e3, 4),
        "value_cold": round(P * n_total * args.steps / dt_cold / 1e6, 2),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": workload_name(W, H, args.kfs, N, args.res, args.independent) +
                        (", i.i.d. uniform u8 noise images" if args.noise else "") +
                        (", depth prior s = %g mu" % args.prior_spread if args.prior_spread != 0.1 else "") +
                        (", foreground strip with occluding edges" if args.scene == "strip" else "") +
                        (", keyframes rolled by up to +-%g degrees (pairs carry their median rotation)" % args.roll
                         if args.roll != 1.0 else "") +
                        (", %d wrong-pose neighbours per keyframe, K1 ONLY stepped (profiling form; `value` is not the "
                         "path's throughput)" % args.outliers if args.outliers else ""),
            "stages": "SemiDenseRecon(K1-K3)+%s+InterKFCheck(K4)+PointSet(K5, back-projected inside K4's kernel)" % xdesc,
            "keyframes_total": n_total, "neighbours": N, "disparity_px": args.disparity,
            "prior_spread": args.prior_spread,
            "parallelism": ("independent x%d" if args.independent else "keyframe-block x%d") % world, "arch": arch,
            "exchange": (args.exchange if exchanging else None),
            "transport": (transport if exchanging else None),
            "rccl_world_size": rccl_world,
            "slots_per_rank": pl["n_slots"],
            "prewarm_steps": n_prewarm,  # untimed steps before the W warm-up steps (clock / cache settling, 0.3 s)
        },
        "stage_ms_per_step": {s: round(v[0] / args.steps, 4) for s, v in timing.items()},
        "roofline": rf,
        "step_roofline": step_roofline(wl, ms_step),
    }
    if exchange_ms:
        out["exchange_ms_per_step"] = exchange_ms
        for other, dt2 in others.items():
            out["value_" + other] = round(P * n_total * args.steps / dt2 / 1e6, 2)
        # which number is which: BASELINE.json words the exchange as "RCCL all-gather of per-KF depth/sigma maps", i.e. every
        # rank's whole block = the allgather_full form; `value` is the optimised variant (only the maps another rank reads,
        # in the compact wire format).  Both are in this line; compare like with like.
        out["value_exchange"] = args.exchange
        out["value_baseline_literal"] = (round(value, 2) if args.exchange == "allgather_full" else
                                         round(P * n_total * args.steps / others["allgather_full"] / 1e6, 2))
        out["exchange_note"] = ("value = %s exchange (%s); value_baseline_literal = value_allgather_full = the all-gather of every "
                                "rank's whole block, BASELINE.json's literal collective.  Multi-GPU figures are unmeasured "
                                "until a driver run on >= 2 GPUs: a failure after transfers are posted is fatal (non-zero "
                                "exit), there is no fall-back from there" %
                                (args.exchange, "compact wire format" if wire_entries > 0 else "whole maps"))
        out["config"]["allgather_full_pieces"] = pkg.shard.AG_PIECES if transport == "native" else 1
        out["config"]["exchange_maps_per_rank"] = {  # maps every rank RECEIVES per step
            "allgather": (world - 1) * pl["contrib_count"], "allgather_late": (world - 1) * pl["contrib_count"],
            "allgather_full": (world - 1) * pl["count"],
            "halo": sum(len(v) for v in pl["recv"].values())}
        out["config"]["exchange_wire"] = (
            {"format": "{rho,sigma} of the keyframe's active-list entries + list length / hash header (sdm_exchange_compact); "
                       "the receiver scatters them through its own list of that keyframe" +
                       (" -- host-staged over gloo in this rehearsal" if rehearse else ""), "entries_per_map": wire_entries,
             "bytes_per_map": 8 * wire_entries, "whole_map_bytes": 8 * P}
            if wire_entries > 0 else {"format": "whole maps", "bytes_per_map": 8 * P})
    if transport_note:
        out["transport_note"] = transport_note
    if rehearse:
        out["rehearsal"] = "all ranks on GPU 0, gloo, host-staged exchange: control-flow dry run, not a measurement"
        out["staged_refused_maps"] = getattr(eng, "staged_refused", 0)
    if stats:
        out["scan"] = scan_record(stats, k1_avg_ms)

    # ---- CPU baseline: the oracle (a port; the reference itself cannot be built) on a bounded sample
    if args.cpu_kfs > 0 and world == 1:  # rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(args, wl)
    out["gen_s"] = round(wl.t_gen, 2)
    if upload_ms is not None:
        # the block's keyframes uploaded in one batch call, then the step
        def incl(ms_kf):
            return round(P * n_total / (dt / args.steps + ms_kf * 1e-3 * len(pl["own"])) / 1e6, 2)
        out["host_upload_ms_per_keyframe"] = round(upload_ms["batch_pageable"], 4)
        out["host_upload"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in upload_ms.items()}
        out["host_upload"]["unit"] = "ms per keyframe, wall clock incl. the device pre-pass (best of 3)"
        if streaming and streaming.get("pageable"):
            # sustained rate with the NEXT block's upload overlapping this block's step (double-buffered slots), everything
            # included; value_pcie_inclusive below is the same work in series
            out["value_streaming"] = round(streaming["pageable"], 2)
            out["value_streaming_pinned"] = round(streaming["pinned"], 2) if streaming.get("pinned") else None
        elif streaming:
            out["value_streaming_error"] = streaming.get("error")
        out["value_pcie_inclusive"] = incl(upload_ms["batch_pageable"])
        out["value_pcie_inclusive_pinned"] = incl(upload_ms["batch_pinned"])
        out["value_pcie_inclusive_per_keyframe_calls"] = incl(upload_ms["per_keyframe_calls"])

    # ---- the other single-GPU BASELINE workloads, each measured like the headline one ----------------------
    if world == 1 and not args.no_extra and (args.res, args.kfs, args.nbrs, args.prior_spread, args.scene, args.roll) == (
            "480p", 64, 20, 0.1, "plane", 1.0):
        wl.close()
        extra = []
        xs = max(5, args.steps // 2)
        for kw in (dict(res="480p", kfs=256, N=20, disparity=args.disparity),                      # north_star target case
                   dict(res="720p", kfs=256, N=7, disparity=args.disparity),                       # BASELINE configs[2]
                   dict(res="480p", kfs=64, N=20, disparity=args.disparity, outliers=2,
                        tag="configs[1] with 2 outlier (wrong-pose) neighbours per keyframe"),
                   dict(res="480p", kfs=64, N=20, disparity=args.disparity, noise=True,
                        tag="configs[1] geometry on i.i.d. uniform u8 images (SURVEY.md §8d adversarial set: scan-only rate)"),
                   dict(res="480p", kfs=64, N=20, disparity=10.0,
                        tag="configs[1] with a long baseline (adjacent-keyframe disparity 10 px)"),
                   dict(res="480p", kfs=64, N=20, disparity=args.disparity, spread=0.3,
                        tag="configs[1] with a wide depth prior (s = 0.3 mu, PM.cc:381-382)"),
                   dict(res="480p", kfs=64, N=20, disparity=10.0, spread=0.3,
                        tag="configs[1] with a long baseline (10 px) AND a wide depth prior (s = 0.3 mu)"),
                   dict(res="480p", kfs=64, N=20, disparity=args.disparity, spread=0.3, strip=True, roll=5.0,
                        tag="configs[1] on the two-plane scene (foreground strip, occluding edges), keyframes rolled by up to "
                            "+-5 degrees with the matching median rotations, depth prior s = 0.3 mu")):
            try:
                extra.append(run_extra(pkg, torch, kw["res"], kw["kfs"], kw["N"], kw["disparity"], xs, 2, local_rank,
                                       barrier, noise=kw.get("noise", False), outliers=kw.get("outliers", 0),
                                       tag=kw.get("tag"), spread=kw.get("spread", 0.1), strip=kw.get("strip", False),
                                       roll=kw.get("roll", 1.0)))
            except Exception as e:  # the headline line must survive a failing extra
                extra.append({"workload": kw.get("tag") or "%s x %d KF x N=%d" % (kw["res"], kw["kfs"], kw["N"]),
                              "error": repr(e)})
        out["extra_configs"] = extra
    else:
        eng.close()
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def streaming_rate(pkg, torch, wl, iters=40, pinned=False):
    """Sustained ingest + compute (frames arrive continuously in the fork: src/Tracking.cc:266-271 -> Modeler.cc:1496-1514):
    the block's keyframes arrive as host gray images every iteration -- sdm_upload_images_batch into the other half of a
    double-sized slot pool, issued BEFORE the current block's step is queued: with the streaming ingest on
    (sdm_set_ingest_overlap: twelve chunk buffers) its staging and H2D copies run while the previous step executes, its
    pre-pass right behind that step, and the list lengths are back long before the block's own step is queued -- the host
    never waits for the GPU, the GPU never for the host.  Returns Mpix*KF/s over `iters` blocks, wall clock, everything
    included; the last blocks' maps are compared with a serial upload-then-step of the same images (bit-equal, or the figure
    is not reported)."""
    import numpy as np
    pl = wl.pl
    ks = [k for k in pl["own"] if k in wl.images]
    if len(ks) != len(pl["own"]):
        return None
    n_slots = pl["n_slots"]
    eng = pkg.Engine(wl.W, wl.H, 2 * n_slots, max_neighbours=wl.N, batch_capacity=min(wl.kfs, 64), with_pointset=True)
    eng.set_ingest_overlap(True)
    ims = [wl.images[k] for k in ks]
    block = None
    if pinned:
        block = eng.host_alloc((len(ks), wl.H, wl.W))
        for i, im in enumerate(ims):
            block[i][...] = im
        ims = [block[i] for i in range(len(ks))]
    poses = [wl.scene.Tcw(k) for k in ks]
    shift = lambda lst, off: [s_ + off for s_ in lst]
    pls = []
    for half in (0, 1):
        off = half * n_slots
        q = dict(pl, own_slots=shift(pl["own_slots"], off), nbr_slots=[shift(r, off) for r in pl["nbr_slots"]])
        if "rot_of" in pl:
            q["rot_of"] = {s_ + off: v for s_, v in pl["rot_of"].items()}
        pls.append(q)
    slots = [[pl["slot"][k] + half * n_slots for k in ks] for half in (0, 1)]

    def step(half):
        pkg.shard.pipeline_step(eng, None, pls[half], wl.min_d, wl.max_d, "none", None, "torch")

    # serial reference: upload, step, read back
    eng.upload_images_batch(slots[0], ims, wl.K, poses)
    step(0)
    probe = (0, len(ks) // 2, len(ks) - 1)
    want = [eng.download_checked(slots[0][i]) for i in probe]
    for _ in range(2):  # warm-up round trips over both halves
        for half in (1, 0):
            eng.upload_images_batch(slots[half], ims, wl.K, poses)
            step(half)
    eng.synchronize()
    t0 = time.perf_counter()
    eng.upload_images_batch(slots[0], ims, wl.K, poses)
    for i in range(iters):
        if i + 1 < iters:  # the NEXT block first: copied while the previous step runs, pre-pass queued ahead of this step
            eng.upload_images_batch(slots[(i + 1) & 1], ims, wl.K, poses)
        step(i & 1)        # asynchronous: returns when the launches are queued (this block's list lengths arrived a step ago)
    eng.synchronize()
    dt = time.perf_counter() - t0
    ok = True
    for half in ((iters - 1) & 1, (iters - 2) & 1):  # the last two blocks: both halves are still intact
        for j, i in enumerate(probe):
            ok = ok and np.array_equal(eng.download_checked(slots[half][i]).view(np.uint32), want[j].view(np.uint32))
    if block is not None:
        eng.host_free(block)
    eng.close()
    if not ok:
        raise RuntimeError("streaming: the overlapped order's maps differ from the serial order's")
    return wl.P * len(ks) * iters / dt / 1e6


def cpu_baseline(args, wl):
    """Times oracle/pm_oracle.c (checker + CPU baseline ONLY; never on the product path) on the
    first cpu_kfs keyframes of the same workload, same stages as the GPU step:
      (i)  1 thread = the reference's effective behaviour (its OpenMP pragmas are inert);
      (ii) all host cores, the rows of ALL sample keyframes shared out at once (pmo_*_batch: collapse over
           (keyframe,row), no Python in the loop) -- what PM.cc:197's pragma intends, at its best."""
    import ctypes
    import platform
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pm_oracle
    eng, scene, images, N, pl = wl.eng, wl.scene, wl.images, wl.N, wl.pl
    out_dir = os.path.join(ROOT, "gpurun_out", "_oracle_build")
    ref = pm_oracle.Oracle("omp", out_dir=out_dir)  # -O3 -march=native -fopenmp, built on THIS box
    ncores = ref.num_threads()
    sample = [k for k in range(args.cpu_kfs) if all(j in images for j in scene.neighbours(k, wl.n_total, N))]
    need = sorted(set(sample) | {j for k in sample for j in scene.neighbours(k, wl.n_total, N)})
    idx = {k: i for i, k in enumerate(need)}
    kfs, gpu_rho, gpu_sig = [], [], []
    for k in need:
        g, t, s = ref.gradient_prepass(images[k])
        kfs.append(ref.keyframe(images[k], g, t, s, scene.K(), scene.Tcw(k)))
        r, sg = eng.download_depth(pl["slot"][k])  # neighbours' finished maps for the inter-keyframe check
        gpu_rho.append(r)
        gpu_sig.append(sg)
    ref_idx = [idx[k] for k in sample]
    nbr_idx = [[idx[j] for j in scene.neighbours(k, wl.n_total, N)] for k in sample]

    def run():
        rho, sigma, st = ref.recon_batch(kfs, ref_idx, nbr_idx, wl.min_d, wl.max_d)
        chk, _ = ref.inter_pointset_batch(kfs, ref_idx, nbr_idx, gpu_rho, gpu_sig, rho)
        return rho, sigma, chk, st

    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(1)
    except OSError:
        omp = None
    t0 = time.perf_counter()
    rho, sigma, chk, st = run()
    t1 = time.perf_counter() - t0
    tn = t1
    if omp is not None:
        omp.omp_set_num_threads(ncores)
        t0 = time.perf_counter()
        run()
        tn = time.perf_counter() - t0
    # live parity check of the GPU result against the oracle on the sample (depth L1 vs ref)
    l1, nmask, mism = 0.0, 0, 0
    for i, k in enumerate(sample):
        for got, want in ((gpu_rho[idx[k]], rho[i]), (eng.download_checked(pl["slot"][k]), chk[i])):
            m = (want > 1e-6)
            mism += int(((got > 1e-6) != m).sum())
            l1 += float(np.abs(got[m] - want[m]).sum())
            nmask += int(m.sum())
    px = wl.W * wl.H * len(sample)
    cpu_model = platform.processor() or ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": round(px / t1 / 1e6, 3), "unit": "Mpix*KF/s", "cores": 1, "kind": "port",
        "sample": "same stages (K1-K5) on the first %d keyframes of the same workload, oracle/pm_oracle.c "
                  "-O3 -march=native, 1 thread (the reference's OpenMP pragmas are inert)" % len(sample),
        "seconds": round(t1, 2),
        "cpu_model": cpu_model,
        "all_cores": {"value": round(px / tn / 1e6, 3), "cores": ncores if omp is not None else 1,
                      "seconds": round(tn, 2),
                      "schedule": "OpenMP dynamic over (keyframe,row) of the whole sample, all stages in C"},
        "mean_candidates_per_search": round(st["candidates"] / max(st["searches"], 1), 3),
        "parity_on_sample": {"mask_mismatches": mism, "depth_L1": (l1 / max(nmask, 1)), "pixels": nmask},
    }


if __name__ == "__main__":
    main()
