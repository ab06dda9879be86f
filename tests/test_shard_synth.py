"""CPU: synthetic-scene generator, keyframe sharding plan, and the N>1 exchange step on gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_neighbour_order(pkg):
    nb = pkg.synth.Scene.neighbours
    assert nb(5, 20, 4) == [6, 4, 7, 3]          # |dk| ascending, +dk before -dk
    assert nb(0, 20, 4) == [1, 2, 3, 4]          # clipped at the sequence start
    assert nb(19, 20, 3) == [18, 17, 16]
    assert nb(1, 20, 4) == [2, 0, 3, 4]
    with pytest.raises(ValueError):
        nb(0, 4, 4)


def test_scene_determinism_and_geometry(pkg):
    synth = pkg.synth
    cam = synth.scaled_intrinsics(synth.TUM1, 80, 60)
    a = synth.Scene(cam, 123)
    b = synth.Scene(cam, 123)
    ia, ga = a.render(3)
    ib, gb = b.render(3)
    assert torch.equal(ia, ib) and torch.equal(ga, gb)
    assert not torch.equal(ia, synth.Scene(cam, 124).render(3)[0])
    # ground-truth inverse depth is ~1 (Z0 = 1) and the pose is a rigid transform
    assert 0.8 < float(ga.mean()) < 1.2
    T = a.Tcw(3)
    R = T[:, :3].astype(np.float64)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6)
    # adjacent keyframes are disparity_px apart at Z0
    C3, C4 = a.pose(3)[1], a.pose(4)[1]
    assert abs((C4[0] - C3[0]) * cam["fx"] - 2.6) < 1e-9
    mn, mx = a.depth_prior()
    assert abs(mn - 1.25) < 1e-6 and abs(mx - 1 / 1.2) < 1e-6  # PM.cc:381-382 with mu = 1, s = 0.1


def test_shard_plan_covers_everything(pkg):
    shard, nb = pkg.shard, pkg.synth.Scene.neighbours
    n_total, world, n = 32, 4, 6
    seen = []
    for r in range(world):
        pl = shard.plan(n_total, world, r, n, nb)
        assert pl["count"] == 8 and pl["first"] == 8 * r
        seen += pl["own"]
        for k, row in zip(pl["own"], pl["nbrs"]):
            assert row == nb(k, n_total, n)
            assert set(row) <= set(pl["inputs"])       # input halo is local
        assert set(pl["own"]) <= set(pl["inputs"])
        # local slots: ascending keyframe order, own block contiguous
        assert [pl["slot"][k] for k in pl["inputs"]] == list(range(pl["n_slots"]))
        assert pl["own_slots"] == list(range(pl["first_slot"], pl["first_slot"] + pl["count"]))
        assert pl["nbr_slots"] == [[pl["slot"][j] for j in row] for row in pl["nbrs"]]
        assert len(pl["inputs"]) <= 8 + n              # halo is at most n/2 each side (+ clipping)
    assert sorted(seen) == list(range(n_total))
    with pytest.raises(ValueError):
        shard.plan(30, 4, 0, n, nb)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, H, W, out, mode="allgather", n_nbr=4):
    sys.path.insert(0, ROOT)
    import sdm_pkg
    pkg = sdm_pkg.load()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    pl = pkg.shard.plan(n_total, world, rank, n_nbr, pkg.synth.Scene.neighbours)
    slot = pl["slot"]
    pool = torch.zeros((pl["n_slots"], H, W, 2), dtype=torch.float32)  # LOCAL slots: own block + halo only
    # stand-in for K1-K3: this rank fills only its own block with a recognisable pattern
    for k in pl["own"]:
        pool[slot[k], :, :, 0] = k + 0.25
        pool[slot[k], :, :, 1] = -(k + 0.5)
    if mode == "allgather_full":
        pkg.shard.allgather_depth(pool, pl)
    elif mode == "allgather":
        pkg.shard.allgather_boundary(pool, pl)
    else:
        pkg.shard.wait_all(pkg.shard.exchange_halo_async(pool, pl))
    ok = pl["n_slots"] <= pl["count"] + n_nbr  # memory per rank does not grow with the world size
    ok = ok and pl["own_slots"] == list(range(pl["first_slot"], pl["first_slot"] + pl["count"]))
    for k in pl["inputs"]:  # every keyframe this rank touches now holds its owner's map, in its slot
        ok = ok and bool((pool[slot[k], :, :, 0] == k + 0.25).all()) and bool((pool[slot[k], :, :, 1] == -(k + 0.5)).all())
    # every neighbour a rank's K4 will read is present
    for row in pl["nbr_slots"]:
        for s in row:
            ok = ok and float(pool[s, 0, 0, 1]) < 0
    out[rank] = ok
    dist.destroy_process_group()


def _kf_list(k, H, W, moved=False):
    """a keyframe's active-pixel list as every rank that holds the keyframe's image derives it: (y << 16 | x), raster
    order, here a deterministic pseudo-random ~25 % of the inset pixels; moved: one pixel displaced (same length)"""
    rng = np.random.default_rng(1000 + k)
    m = rng.random((H, W)) < 0.25
    m[:2] = m[-2:] = False
    m[:, :2] = m[:, -2:] = False
    if moved:
        ys, xs = np.nonzero(m)
        m[ys[3], xs[3]] = False
        ys0, xs0 = np.nonzero(~m[2:-2, 2:-2])
        m[ys0[5] + 2, xs0[5] + 2] = True
    ys, xs = np.nonzero(m)
    return (ys.astype(np.uint32) << 16) | xs.astype(np.uint32)


def _kf_map(k, lst, H, W):
    m = np.zeros((H, W, 2), np.float32)
    t = np.arange(lst.size, dtype=np.float32)
    m[lst >> 16, lst & 0xFFFF, 0] = k + 0.25 + t / 4096
    m[lst >> 16, lst & 0xFFFF, 1] = -(k + 0.5) - t / 8192
    return m


def _compact_worker(rank, world, port, n_total, H, W, out, mode, n_nbr):
    """the compact wire format between PROCESSES (gloo): every rank packs the maps it owns through its own copy of the
    keyframe's list, the receiver scatters them through ITS copy; a receiver whose list differs refuses the map"""
    sys.path.insert(0, ROOT)
    import sdm_pkg
    pkg = sdm_pkg.load()
    shard = pkg.shard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    pl = shard.plan(n_total, world, rank, n_nbr, pkg.synth.Scene.neighbours)
    slot = pl["slot"]
    odd = sorted(j for v in pl["recv"].values() for j in v)[0] if (mode.endswith("refuse") and rank == 1) else None
    lists = {slot[k]: _kf_list(k, H, W, moved=(k == odd)) for k in pl["inputs"]}
    # the wire format of the job: the longest list anywhere, rounded up to 64 (one all-reduce, as agree_compact_wire does)
    t = torch.tensor([max(v.size for v in lists.values())], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    entries = (int(t.item()) + 63) // 64 * 64
    pool = torch.zeros((pl["n_slots"], H, W, 2), dtype=torch.float32)
    for k in pl["own"]:
        pool[slot[k]] = torch.from_numpy(_kf_map(k, lists[slot[k]], H, W))
    codec = shard.HostCompactCodec(pool, lists, entries)
    if mode.startswith("allgather"):
        shard.allgather_boundary_compact(codec, pl)
    else:
        shard.exchange_halo_compact(codec, pl)
    ok = entries * 2 < H * W  # the payload is smaller than the plane it stands for
    for k in pl["inputs"]:
        want = _kf_map(k, _kf_list(k, H, W), H, W)
        if k == odd:  # refused: counted, and the plane keeps what it held (zeros)
            ok = ok and codec.refused == 1 and not bool(pool[slot[k]].any())
        else:
            ok = ok and bool((pool[slot[k]].numpy().view(np.uint32) == want.view(np.uint32)).all())
    if odd is None:
        ok = ok and codec.refused == 0
    out[rank] = ok
    dist.destroy_process_group()


def _run_world(world, n_total, mode, n_nbr=4):
    if "compact" in mode:
        return _run_world_with(_compact_worker, world, n_total, mode, n_nbr, 24, 32)
    return _run_world_with(_worker, world, n_total, mode, n_nbr, 6, 8)


def _run_world_with(target, world, n_total, mode, n_nbr, H, W):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=target, args=(r, world, port, n_total, H, W, out, mode, n_nbr)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert dict(out) == {r: True for r in range(world)}


def test_halo_plan_is_pairwise_consistent(pkg):
    shard, nb = pkg.shard, pkg.synth.Scene.neighbours
    for (n_total, world, n) in [(32, 4, 6), (64, 8, 20), (24, 2, 7), (512, 8, 20)]:
        plans = [shard.plan(n_total, world, r, n, nb) for r in range(world)]
        for r, pl in enumerate(plans):
            assert sorted(pl["boundary"] + pl["interior"]) == pl["own"]
            for q, lst in pl["recv"].items():
                assert plans[q]["send"][r] == lst, "what r receives from q is what q sends to r"
            for q, lst in pl["send"].items():
                assert plans[q]["recv"][r] == lst
            got = set(pl["own"]) | {j for lst in pl["recv"].values() for j in lst}
            assert got == set(pl["inputs"])
            send, recv = shard.halo_lists(pl)
            assert len(send) == sum(len(v) for v in pl["send"].values())
            assert sorted(s for _, s in recv) == sorted(pl["slot"][k] for k in pl["inputs"] if k not in pl["own"])
            assert shard.fetch_list(pl) == [(k, pl["slot"][k]) for k in pl["inputs"] if k not in pl["own"]]
        # index-local covisibility: only adjacent blocks talk, N/2 keyframes each way
        mid = plans[world // 2] if (world > 2 and n_total // world >= n // 2) else None
        if mid is not None:
            assert set(mid["recv"]) == {world // 2 - 1, world // 2 + 1}
            assert all(len(v) == n // 2 for v in mid["recv"].values())


def test_halo_exchange_gloo_world2(pkg):
    """the point-to-point halo exchange between K3 and K4, world_size 2 on gloo"""
    _run_world(2, 12, "halo")


def test_halo_exchange_gloo_world3(pkg):
    """a middle rank exchanges with both neighbours at once (batched isend/irecv)"""
    _run_world(3, 24, "halo", n_nbr=6)


def test_allgather_exchange_gloo_world2(pkg):
    """the whole-block all-gather form of the exchange, world_size 2 on gloo"""
    _run_world(2, 12, "allgather_full")


def test_boundary_allgather_gloo_world3(pkg):
    """the default exchange: an all-gather of the boundary keyframes only (padded to a common count: the end ranks
    have half as many), world_size 3 on gloo -- every map a rank's K4 reads arrives in its local slot"""
    _run_world(3, 24, "allgather", n_nbr=6)


def test_compact_wire_gloo_world2(pkg):
    """the compact wire format ({rho,sigma} of the list entries + length / hash header) crosses a process boundary:
    point-to-point form, world_size 2 on gloo, host arrays (the numpy statement of k_pack_lists / k_unpack_lists)"""
    _run_world(2, 12, "halo_compact")


def test_compact_wire_gloo_world3_allgather(pkg):
    _run_world(3, 24, "allgather_compact", n_nbr=6)


def test_compact_wire_refuses_a_different_list(pkg):
    """a receiver whose list of a keyframe differs from the sender's -- SAME length, one pixel moved: only the hash can
    tell -- refuses that map, counts it and leaves the plane alone"""
    _run_world(2, 12, "halo_compact_refuse")


def test_compact_codec_roundtrip(pkg):
    shard = pkg.shard
    H, W = 24, 32
    lst = _kf_list(7, H, W)
    m = _kf_map(7, lst, H, W)
    entries = (lst.size + 63) // 64 * 64
    p = shard.pack_compact(m, lst, entries)
    assert p.shape == (entries + shard.XCHG_HEADER, 2) and int(p.view(np.uint32)[entries, 0]) == lst.size
    back = np.zeros_like(m)
    assert shard.unpack_compact(p, lst, entries, back) and (back.view(np.uint32) == m.view(np.uint32)).all()
    other = _kf_list(7, H, W, moved=True)
    assert other.size == lst.size and shard.list_hash(other) != shard.list_hash(lst)
    untouched = np.zeros_like(m)
    assert not shard.unpack_compact(p, other, entries, untouched) and not untouched.any()
    with pytest.raises(ValueError):
        shard.pack_compact(m, lst, 64)


def test_boundary_allgather_lists(pkg):
    """contribution / fetch lists of the boundary all-gather: the same contribution table on every rank, every fetched
    position holds the keyframe the plan says, padded contributions repeat a real slot"""
    shard, nb = pkg.shard, pkg.synth.Scene.neighbours
    for (n_total, world, n) in [(32, 4, 6), (64, 8, 20), (512, 8, 20), (24, 2, 7), (16, 4, 2)]:
        plans = [shard.plan(n_total, world, r, n, nb) for r in range(world)]
        for r, pl in enumerate(plans):
            assert pl["contrib"] == plans[0]["contrib"] and pl["contrib_count"] == plans[0]["contrib_count"]
            assert pl["contrib"][r] == pl["boundary"]
            cs = shard.contrib_slots(pl)
            assert len(cs) == pl["contrib_count"] and set(cs) <= set(pl["own_slots"])
            assert cs[:len(pl["boundary"])] == pl["boundary_slots"]
            fetched = {}
            for idx, s in shard.contrib_fetch_list(pl):
                q, pos = divmod(idx, pl["contrib_count"])
                fetched[s] = pl["contrib"][q][pos]
            want = {pl["slot"][k]: k for k in pl["inputs"] if k not in pl["own"]}
            assert fetched == want
        # the collective moves (world - 1) * contrib_count maps per rank instead of (world - 1) * block
        assert plans[0]["contrib_count"] <= min(n, n_total // world)
    assert shard.sub_blocks(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)] and shard.sub_blocks(3, 8) == [(0, 1), (1, 1), (2, 1)]


def test_list_hash_definition(pkg):
    """shard.list_hash (numpy) against the definition spelled out with Python integers (csrc/sdm_ingest.h seg_hash_term): per
    64-pixel row segment with listed pixels, SplitMix64's finaliser of mask ^ ((y << 16 | x0) * 0x9E3779B97F4A7C15), summed
    mod 2^64; order-independent; sensitive to one pixel moving inside a segment, across segments and across rows; a known
    answer pins the constants."""
    shard = pkg.shard
    M = (1 << 64) - 1

    def by_definition(lst):
        segs = {}
        for v in np.asarray(lst, np.uint32).tolist():
            segs[v & ~63] = segs.get(v & ~63, 0) | (1 << (v & 63))
        tot = 0
        for key, mask in segs.items():
            z = mask ^ ((key * 0x9E3779B97F4A7C15) & M)
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
            tot = (tot + (z ^ (z >> 31))) & M
        return tot

    rng = np.random.default_rng(11)
    assert shard.list_hash(np.zeros(0, np.uint32)) == 0
    assert shard.list_hash(np.uint32([(2 << 16) | 2])) == by_definition([(2 << 16) | 2]) == 0xC6AC4D8F8B52AE06  # known answer
    for n in (1, 2, 63, 64, 65, 1000, 50000):
        ys, xs = rng.integers(2, 478, n), rng.integers(2, 638, n)
        lst = np.unique((ys.astype(np.uint32) << 16) | xs.astype(np.uint32))
        h = shard.list_hash(lst)
        assert h == by_definition(lst)
        assert shard.list_hash(lst[::-1].copy()) == h  # a set hash
        moved = lst.copy()
        moved[0] ^= 1  # the neighbouring column (same segment)
        if moved[0] not in lst[1:]:
            assert shard.list_hash(moved) != h
        moved = lst.copy()
        moved[0] ^= 64  # the same bit of the neighbouring segment
        if moved[0] not in lst[1:]:
            assert shard.list_hash(moved) != h
        moved = lst.copy()
        moved[0] ^= 1 << 16  # the neighbouring row
        if moved[0] not in lst[1:]:
            assert shard.list_hash(moved) != h
