"""GPU: the native exchange entry points of the C ABI (include/sdm_c.h sdm_comm_* / sdm_exchange_* /
sdm_allgather_depth) at world size 1, where they must be exact no-ops, plus their argument checking.  The RCCL
transport BETWEEN GPUs cannot run on a one-GPU box (RCCL refuses two ranks on one device): unmeasured on hardware
until the driver's multi-GPU run.  What can run here does: test_rccl_single_rank_rehearsal drives every RCCL call
of sdm_comm.h through a real one-rank communicator (SDM_COMM_SINGLE_RANK_RCCL=1); the multi-rank control flow is
covered with the torch/gloo transport in test_gpu_shard.py and the plan/list logic on CPU in test_shard_synth.py."""
import numpy as np
import pytest

from common import Sequence, assert_bit_equal

pytestmark = pytest.mark.gpu


def test_world1_exchange_is_a_noop(pkg, oracle, gpu_ok):
    seq = Sequence(pkg, oracle, 96, 72, 8, 0x5EED0C11)
    n = 5
    eng = pkg.Engine(seq.W, seq.H, seq.n_kf, max_neighbours=n)
    seq.upload(eng, device_prepass=True)
    assert eng.comm_info() == (1, 0)
    eng.comm_init(None, 1, 0)  # world size 1 needs no unique id and never loads RCCL
    assert eng.comm_info() == (1, 0)
    pl = pkg.shard.plan(seq.n_kf, 1, 0, n, seq.scene.neighbours)
    assert pl["n_slots"] == seq.n_kf and pkg.shard.halo_lists(pl) == ([], []) and pkg.shard.fetch_list(pl) == []
    refs, nbrs = pl["own_slots"], pl["nbr_slots"]
    eng.recon(refs, nbrs, seq.min_depth, seq.max_depth)
    before = [eng.download_depth(k) for k in refs]
    eng.exchange_halo_begin([], [])
    eng.exchange_wait()
    eng.exchange_halo([], [])
    eng.allgather_depth(0, seq.n_kf, fetch=[])      # gathered-buffer form
    eng.allgather_depth(0, seq.n_kf, fetch=None)    # in-place form (slot == global keyframe index)
    for k, (r, s) in zip(refs, before):
        g = eng.download_depth(k)
        assert_bit_equal(g[0], r)
        assert_bit_equal(g[1], s)
    # the whole step through the native transport == the plain calls
    for exch in ("halo", "allgather", "allgather_full"):
        pkg.shard.pipeline_step(eng, None, pl, seq.min_depth, seq.max_depth, exch, transport="native")
        chk = [eng.download_checked(k) for k in refs]
        eng.recon(refs, nbrs, seq.min_depth, seq.max_depth)
        eng.inter_check(refs, nbrs)
        for k in refs:
            assert_bit_equal(eng.download_checked(k), chk[k], "checked rho kf %d (%s)" % (k, exch))
    # the overlapped all-gather steps at world size 1 (the pieces are device copies; schedule and results are the plain
    # step's): the boundary form (nothing crosses ranks here: a one-map padded contribution) and the whole-block form
    # in 2, 3 and 8 sub-blocks
    want = [eng.download_checked(k) for k in refs]
    for exch, pieces in (("allgather", 1), ("allgather_full", 2), ("allgather_full", 3), ("allgather_full", 8)):
        pkg.shard.pipeline_step(eng, None, pl, seq.min_depth, seq.max_depth, exch, transport="native",
                                ag_pieces=pieces, force_pieces=True)
        for k in refs:
            assert_bit_equal(eng.download_checked(k), want[k], "checked rho kf %d (%s, %d pieces)" % (k, exch, pieces))
    eng.comm_destroy()
    eng.close()


def test_allgather_pieces_addressing(pkg, oracle, gpu_ok):
    """sdm_allgather_begin / _piece / _finish at world size 1: every piece lands in the gather buffer -- straight from
    the pool for a run of consecutive slots, through the packing buffer for any other list -- and fetch_index =
    owner * maps_per_rank + position finds it again (the same code addresses the other ranks' maps)"""
    seq = Sequence(pkg, oracle, 96, 72, 8, 0x5EED0C12)
    n = 5
    eng = pkg.Engine(seq.W, seq.H, seq.n_kf + 4, max_neighbours=n)
    seq.upload(eng, device_prepass=True)
    refs = list(range(seq.n_kf))
    nbrs = [seq.neighbours(k, n) for k in refs]
    with pytest.raises(pkg.SdmError) as e:
        eng.allgather_piece([0, 1])  # not begun
    assert e.value.code == 4
    eng.allgather_begin(seq.n_kf)
    with pytest.raises(pkg.SdmError) as e:
        eng.allgather_begin(seq.n_kf)  # already open
    assert e.value.code == 4
    with pytest.raises(pkg.SdmError) as e:
        eng.allgather_piece([0, 1, 2])  # no depth maps yet
    assert e.value.code == 4
    order = [[0, 1, 2], [5, 3, 4], [7, 6]]  # a run, a permuted list (packed), a reversed pair (packed)
    for i, piece in enumerate(order):
        eng.recon(piece, [nbrs[k] for k in piece], seq.min_depth, seq.max_depth)
        eng.allgather_piece(piece)
        if i == 0:
            with pytest.raises(pkg.SdmError) as e:
                eng.allgather_finish([])  # the pieces do not add up yet
            assert e.value.code == 4
    with pytest.raises(pkg.SdmError) as e:
        eng.allgather_piece([0])  # more than announced
    assert e.value.code == 1
    maps = {k: eng.download_depth(k) for k in refs}
    with pytest.raises(pkg.SdmError) as e:
        eng.allgather_finish([(5, 2)])  # would overwrite a contributed map
    assert e.value.code == 1
    with pytest.raises(pkg.SdmError) as e:
        eng.allgather_finish([(5, 9), (6, 9)])  # duplicate destination
    assert e.value.code == 1
    flat = [k for piece in order for k in piece]  # position -> keyframe
    fetch = [(3, 8), (0, 9), (6, 10), (5, 11)]
    eng.allgather_finish(fetch)
    for pos, s in fetch:
        r, sg = eng.download_depth(s)
        assert_bit_equal(r, maps[flat[pos]][0], "fetched rho at position %d (keyframe %d)" % (pos, flat[pos]))
        assert_bit_equal(sg, maps[flat[pos]][1], "fetched sigma at position %d" % pos)
    assert float(np.abs(maps[5][0]).sum()) > 0
    assert eng.comm_all_ok(True) is True and eng.comm_all_ok(False) is False  # world size 1: the local verdict
    assert eng.comm_all_max(77) == 77
    eng.close()


def test_rccl_single_rank_rehearsal(pkg, oracle, gpu_ok, monkeypatch):
    """Every RCCL call the exchange makes -- ncclGetUniqueId, ncclCommInitRank, ncclAllGather (from the pool, from the
    packing buffer, in place, on the exchange stream and on the compute stream), grouped ncclSend/ncclRecv, ncclAllReduce,
    ncclCommDestroy -- through a real communicator of ONE rank: librccl is loaded, the kernels run on this GPU, the
    stream/event ordering is the multi-rank one, and every map that went through RCCL is bit-identical to its source."""
    monkeypatch.setenv("SDM_COMM_SINGLE_RANK_RCCL", "1")
    seq = Sequence(pkg, oracle, 96, 72, 8, 0x5EED0C13)
    n = 5
    eng = pkg.Engine(seq.W, seq.H, seq.n_kf + 6, max_neighbours=n)
    seq.upload(eng, device_prepass=True)
    try:
        eng.comm_init(eng.comm_unique_id(), 1, 0)
    except pkg.SdmError as e:  # no librccl / no bootstrap interface on this box: an environment matter, not a result
        eng.close()
        if e.code != 5:  # SDM_ECOMM
            raise
        pytest.skip("RCCL communicator cannot be created here: %s" % e)
    assert eng.comm_info() == (1, 0)
    assert eng.comm_all_ok(True) is True and eng.comm_all_ok(False) is False  # ncclAllReduce(min)
    assert eng.comm_all_max(12345) == 12345 and eng.comm_all_max(0) == 0        # ncclAllReduce(max)
    refs = list(range(seq.n_kf))
    nbrs = [seq.neighbours(k, n) for k in refs]
    # pieces gathered while the next keyframes are reconstructed
    eng.allgather_begin(seq.n_kf)
    order = [[0, 1, 2], [5, 3, 4], [7, 6]]
    for piece in order:
        eng.recon(piece, [nbrs[k] for k in piece], seq.min_depth, seq.max_depth)
        eng.allgather_piece(piece)
    flat = [k for piece in order for k in piece]
    fetch = [(3, 8), (0, 9), (6, 10), (5, 11)]
    eng.allgather_finish(fetch)
    maps = {k: eng.download_depth(k) for k in refs}
    assert float(np.abs(maps[5][0]).sum()) > 0
    for pos, s in fetch:
        r, sg = eng.download_depth(s)
        assert_bit_equal(r, maps[flat[pos]][0], "RCCL-gathered rho at position %d" % pos)
        assert_bit_equal(sg, maps[flat[pos]][1], "RCCL-gathered sigma at position %d" % pos)
    # the one-call forms: into the gather buffer + fetch, and in place
    eng.allgather_depth(0, seq.n_kf, fetch=[(2, 12), (7, 13)])
    for pos, s in ((2, 12), (7, 13)):
        r, sg = eng.download_depth(s)
        assert_bit_equal(r, maps[pos][0])
        assert_bit_equal(sg, maps[pos][1])
    eng2 = pkg.Engine(seq.W, seq.H, seq.n_kf, max_neighbours=n)
    seq.upload(eng2, device_prepass=True)
    eng2.comm_init(eng2.comm_unique_id(), 1, 0)
    eng2.recon(refs, nbrs, seq.min_depth, seq.max_depth)
    eng2.allgather_depth(0, seq.n_kf, fetch=None)
    for k in refs:
        r, sg = eng2.download_depth(k)
        assert_bit_equal(r, maps[k][0], "in-place all-gather kf %d" % k)
        assert_bit_equal(sg, maps[k][1])
    eng2.comm_destroy()
    eng2.close()
    # grouped send / recv on the exchange stream (to itself: the only peer there is), overlapped with more compute
    eng.exchange_halo_begin([(0, 1), (0, 6)], [(0, 8), (0, 9)])
    eng.recon([2, 3], [nbrs[2], nbrs[3]], seq.min_depth, seq.max_depth)
    eng.exchange_wait()
    for src, dst in ((1, 8), (6, 9)):
        r, sg = eng.download_depth(dst)
        assert_bit_equal(r, maps[src][0], "self send/recv slot %d -> %d" % (src, dst))
        assert_bit_equal(sg, maps[src][1])
    # the three step forms of shard.pipeline_step over this communicator == the plain calls
    pl = pkg.shard.plan(seq.n_kf, 1, 0, n, seq.scene.neighbours)
    eng.recon(pl["own_slots"], pl["nbr_slots"], seq.min_depth, seq.max_depth)
    eng.inter_check(pl["own_slots"], pl["nbr_slots"])
    want = [eng.download_checked(k) for k in refs]
    assert float(np.abs(want[4]).sum()) > 0
    for wire in ("whole", "compact"):
        entries = pkg.shard.agree_compact_wire(eng, pl) if wire == "compact" else 0
        assert (entries > 0) == (wire == "compact")
        for exch, pieces in (("halo", 1), ("allgather", 1), ("allgather_late", 1), ("allgather_full", 3)):
            pkg.shard.pipeline_step(eng, None, pl, seq.min_depth, seq.max_depth, exch, transport="native",
                                    ag_pieces=pieces, force_pieces=True)
            for k in refs:
                assert_bit_equal(eng.download_checked(k), want[k], "checked rho kf %d (%s over RCCL, %s)" % (k, exch, wire))
    eng.exchange_compact(0)
    eng.comm_destroy()
    eng.close()


@pytest.mark.parametrize("rccl", [False, True])
def test_compact_wire_format(pkg, oracle, gpu_ok, monkeypatch, rccl):
    """sdm_exchange_compact: maps cross ranks as the {rho,sigma} of their keyframe's active-list entries and are scattered
    through the RECEIVER's list of the same keyframe (its input halo holds the image).  One GPU: keyframe k is resident
    twice (slots k and 8+k, as on two ranks); every map that went through pack -> gather -> unpack (device-copy stand-in,
    or a real one-rank RCCL communicator: all-gather pieces and grouped send/recv) equals its source bit for bit -- also
    into a destination plane that held an arbitrary map before."""
    if rccl:
        monkeypatch.setenv("SDM_COMM_SINGLE_RANK_RCCL", "1")
    seq = Sequence(pkg, oracle, 96, 72, 8, 0x5EED0C14)
    n, K = 5, seq.n_kf
    eng = pkg.Engine(seq.W, seq.H, 2 * K, max_neighbours=n)
    seq.upload(eng, device_prepass=True)
    for k in range(K):  # the same keyframes again, as another rank's input halo would hold them
        eng.upload_image(K + k, seq.im[k], seq.K, seq.Tcw[k])
    if rccl:
        try:
            eng.comm_init(eng.comm_unique_id(), 1, 0)
        except pkg.SdmError as e:
            eng.close()
            if e.code != 5:
                raise
            pytest.skip("RCCL communicator cannot be created here: %s" % e)
    refs = list(range(K))
    nbrs = [seq.neighbours(k, n) for k in refs]
    eng.recon(refs, nbrs, seq.min_depth, seq.max_depth)
    maps = {k: eng.download_depth(k) for k in refs}
    counts = [eng.active_count(k) for k in range(2 * K)]
    assert counts[:K] == counts[K:] and max(counts) > 100
    E = (max(counts) + 63) // 64 * 64
    assert E < seq.W * seq.H // 2
    with pytest.raises(pkg.SdmError) as e:
        eng.exchange_compact(seq.W * seq.H + 1)
    assert e.value.code == 1
    eng.exchange_compact(min(counts) - 1)  # too short for at least one list
    eng.allgather_begin(K)
    with pytest.raises(pkg.SdmError) as e:
        eng.allgather_piece(refs)
    assert e.value.code == 4
    eng.close()

    eng = pkg.Engine(seq.W, seq.H, 2 * K, max_neighbours=n)
    seq.upload(eng, device_prepass=True)
    for k in range(K):
        eng.upload_image(K + k, seq.im[k], seq.K, seq.Tcw[k])
    if rccl:
        eng.comm_init(eng.comm_unique_id(), 1, 0)
    eng.recon(refs, nbrs, seq.min_depth, seq.max_depth)
    eng.exchange_compact(E)
    with pytest.raises(pkg.SdmError) as e:  # the one-shot form moves whole maps only
        eng.allgather_depth(0, K, fetch=[])
    assert e.value.code == 4
    rng = np.random.default_rng(5)
    junk = rng.random((seq.H, seq.W), dtype=np.float32)
    eng.upload_depth(K + 3, junk, junk)  # an arbitrary map in a destination plane: zeroed before the scatter
    order = [[0, 1, 2], [5, 3, 4], [7, 6]]
    flat = [k for piece in order for k in piece]
    eng.allgather_begin(K)
    for piece in order:
        eng.allgather_piece(piece)
    eng.allgather_finish([(pos, K + flat[pos]) for pos in range(K)])
    for k in refs:
        r, sg = eng.download_depth(K + k)
        assert_bit_equal(r, maps[k][0], "compact all-gather rho kf %d" % k)
        assert_bit_equal(sg, maps[k][1], "compact all-gather sigma kf %d" % k)
    assert float(np.abs(maps[5][0]).sum()) > 0
    assert eng.exchange_mismatches() == 0
    # K4 reads the received copies like the originals
    eng.inter_check([4], [nbrs[4]])
    want = eng.download_checked(4)
    eng.inter_check([4], [[K + j for j in nbrs[4]]])
    assert_bit_equal(eng.download_checked(4), want, "K4 against compact-received maps")
    if rccl:  # grouped send / recv in compact form (to itself), into planes that hold the previous maps
        eng.recon([1, 6], [nbrs[1], nbrs[6]], seq.min_depth, seq.max_depth)
        eng.exchange_halo([(0, 1), (0, 6)], [(0, K + 1), (0, K + 6)])
        for k in (1, 6):
            r, sg = eng.download_depth(K + k)
            assert_bit_equal(r, maps[k][0], "compact send/recv rho kf %d" % k)
            assert_bit_equal(sg, maps[k][1])
        eng.comm_destroy()  # also back to whole maps: the wire format belongs to the communicator's agreement
        eng.exchange_compact(E)
    # a receiver whose image of the keyframe differs from the sender's (another list) refuses the map and says so
    eng.upload_image(K + 2, seq.im[5], seq.K, seq.Tcw[5])
    assert eng.active_count(K + 2) != eng.active_count(2)
    before = eng.download_depth(K + 2)
    eng.allgather_begin(1)
    eng.allgather_piece([2])
    eng.allgather_finish([(0, K + 2)])
    assert eng.exchange_mismatches() == 1 and eng.exchange_mismatches() == 0
    assert_bit_equal(eng.download_depth(K + 2)[0], before[0], "a refused map leaves the destination untouched")
    # the same format through host memory (sdm_compact_pack_host / _unpack_host), against its numpy statement
    # (shard.pack_compact / unpack_compact -- what the CPU tests move between processes): byte for byte
    shard = pkg.shard
    for k in (0, 5):
        lst, h = eng.active_list(k)
        assert h == shard.list_hash(lst)
        payload = eng.compact_pack_host(k)
        m = np.stack(maps[k], axis=2)
        want = shard.pack_compact(m, lst, E)
        assert (payload.view(np.uint32)[:lst.size] == want.view(np.uint32)[:lst.size]).all()
        assert (payload.view(np.uint32)[E:E + 2] == want.view(np.uint32)[E:E + 2]).all(), "header: length and hash"
        eng.upload_depth(K + k, junk, junk)
        assert not eng.compact_unpack_host(K + k, want)  # a numpy-packed payload, scattered by the device
        r, sg = eng.download_depth(K + k)
        assert_bit_equal(r, maps[k][0], "host payload rho kf %d" % k)
        assert_bit_equal(sg, maps[k][1])
        back = np.zeros_like(m)
        assert shard.unpack_compact(payload, lst, E, back) and (back.view(np.uint32) == m.view(np.uint32)).all()
        # a payload packed with a list of the SAME length but another hash (one bit of the hash flipped) is refused
        forged = payload.copy()
        forged.view(np.uint32)[E, 1] ^= 1
        before = eng.download_depth(K + k)
        assert eng.compact_unpack_host(K + k, forged)
        assert_bit_equal(eng.download_depth(K + k)[0], before[0], "a refused payload leaves the destination untouched")
    assert eng.exchange_mismatches() == 2
    eng.upload_depth(K + 3, junk, junk)  # an arbitrary map is no compact source; a reconstructed or compact-received one is
    assert eng.compact_sources_ready([0, 5, K + 5]) and not eng.compact_sources_ready([0, K + 3])
    eng.exchange_compact(0)
    eng.allgather_depth(0, K, fetch=[])  # whole maps again
    eng.close()


@pytest.mark.parametrize("W,H,n_maps", [(9, 11, 5), (16, 9, 40), (33, 31, 3)])
def test_map_copies_odd_and_many(pkg, gpu_ok, W, H, n_maps):
    """the packing / fetch copies of the all-gather (k_copy_maps: 16-byte units, <= 32 maps per launch) with an odd pixel
    count (plain copies), more maps than one launch takes, and non-contiguous lists -- uploaded random maps, bit-equal"""
    rng = np.random.default_rng(W * 1000 + H)
    eng = pkg.Engine(W, H, 2 * n_maps + 1, max_neighbours=3)
    maps = {}
    src = [2 * i + 1 for i in range(n_maps)]  # every other slot: never a run -> packed through the staging buffer
    for s_ in src:
        maps[s_] = (rng.random((H, W), dtype=np.float32), rng.random((H, W), dtype=np.float32))
        eng.upload_depth(s_, *maps[s_])
    eng.allgather_begin(n_maps)
    eng.allgather_piece(src)
    dst = [2 * i for i in range(n_maps)]
    perm = list(rng.permutation(n_maps))
    eng.allgather_finish([(int(perm[i]), dst[i]) for i in range(n_maps)])
    for i in range(n_maps):
        r, sg = eng.download_depth(dst[i])
        assert_bit_equal(r, maps[src[perm[i]]][0], "rho of map %d" % i)
        assert_bit_equal(sg, maps[src[perm[i]]][1], "sigma of map %d" % i)
    for s_ in src:  # the sources are untouched
        r, sg = eng.download_depth(s_)
        assert_bit_equal(r, maps[s_][0])
    eng.close()


def test_exchange_argument_checks(pkg, gpu_ok):
    eng = pkg.Engine(64, 48, 4, max_neighbours=3)
    with pytest.raises(pkg.SdmError) as e:  # world size 1 has no peers
        eng.exchange_halo_begin([(1, 0)], [])
    assert e.value.code == 1
    with pytest.raises(pkg.SdmError) as e:  # block beyond the slots
        eng.allgather_depth(2, 3, fetch=[])
    assert e.value.code == 1
    with pytest.raises(pkg.SdmError) as e:  # block slots hold no depth map yet
        eng.allgather_depth(0, 2, fetch=[])
    assert e.value.code == 4
    with pytest.raises(pkg.SdmError) as e:
        eng.comm_init(None, 2, 5)
    assert e.value.code == 1
    with pytest.raises(pkg.SdmError) as e:  # a multi-rank communicator needs an id
        eng.comm_init(None, 2, 0)
    assert e.value.code == 1
    eng.mark_depth_present([0, 1])
    eng.allgather_depth(0, 2, fetch=[])
    with pytest.raises(pkg.SdmError):
        eng.mark_depth_present([7])
    eng.close()
