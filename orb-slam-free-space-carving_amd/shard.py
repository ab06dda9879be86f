"""Keyframe sharding for the multi-GPU path.

K1-K3 (SemiDenseRecon, PM.cc:137-256) of keyframe k read only immutable inputs of k and its
covisible neighbours; K4 (InterKeyFrameDepthChecking, PM.cc:628-799) needs the neighbours'
FINISHED {rho, sigma} maps (the reference gates on that at PM.cc:292-298).  So keyframes shard as
contiguous blocks, one block per GPU, with exactly one exchange step between K3 and K4.

Slots are LOCAL: a rank's engine holds only the keyframes it touches -- its own block plus the
input halo (the covisible neighbours that live on adjacent ranks) -- in ascending keyframe order, so
the own block is a contiguous run of slots and device memory per rank does not grow with the world
size.  plan()["slot"] maps a global keyframe index to the local slot.

Three exchange forms, two transports:

  allgather       (default) an RCCL all-gather of the per-keyframe {rho,sigma} maps that cross ranks: every rank
                  contributes its BOUNDARY keyframes -- the own keyframes some other rank's K4 reads -- padded to a
                  common count, right after reconstructing them; the interior keyframes are reconstructed while the
                  collective runs on the engine's exchange stream.  (world-1) x boundary x 8P bytes arrive per rank.
  allgather_full  every rank's WHOLE block (BASELINE.json's literal wording), pipelined in sub-blocks behind the
                  reconstruction; (world-1) x block x 8P bytes per rank -- three times the bytes of the default on the
                  bench's index-local covisibility graph, for maps nobody reads.
  halo            point-to-point: each rank receives only the maps its K4 will read (N/2 keyframes from each adjacent
                  block), 2 x (N/2) x 8P bytes per rank over two direct xGMI links, independent of the number of GPUs,
                  issued right after the boundary keyframes are reconstructed.

  native          RCCL called by the engine itself (include/sdm_c.h sdm_exchange_* / sdm_allgather_depth):
                  the C++ drop-in shards without Python; this module only hands over the lists.
  torch           torch.distributed on the pool tensor (backend nccl = RCCL; gloo with host staging
                  exists to rehearse the control flow with several ranks on ONE GPU or on CPU).
"""
import torch
import torch.distributed as dist


def block_partition(n_total, world, rank):
    """Contiguous equal blocks; n_total must divide evenly (weak scaling: fixed work per GPU)."""
    if n_total % world:
        raise ValueError("n_total (%d) must be a multiple of world size (%d)" % (n_total, world))
    count = n_total // world
    return rank * count, count


def owner_of(k, n_total, world):
    return k // (n_total // world)


def plan(n_total, world, rank, n_nbr, neighbours_fn):
    """Returns a dict:

    first, count  this rank's block of GLOBAL keyframe indices
    own, nbrs     the block and each keyframe's covisible neighbours (global indices)
    inputs        keyframes whose IMAGES this rank must hold = own block + its neighbours, ascending
    slot          {global keyframe: local slot} over `inputs`; n_slots = len(inputs); first_slot
    boundary      own keyframes some OTHER rank's K4 reads (reconstruct these first, then exchange)
    interior      the rest of the own block
    recv          {peer: ascending keyframes owned by peer that this rank's K4 reads}
    send          {peer: ascending own keyframes that peer's K4 reads}
    own_slots, nbr_slots, boundary_slots, interior_slots   the same lists in local slots
    Every rank derives every other rank's needs from the same deterministic neighbour function, so
    send/recv lists match pairwise (k-th send to a peer = that peer's k-th receive) without negotiation."""
    first, count = block_partition(n_total, world, rank)
    own = list(range(first, first + count))
    nbrs = [list(neighbours_fn(k, n_total, n_nbr)) for k in own]
    need = set(own)
    for row in nbrs:
        need.update(row)
    recv, send = {}, {}
    for j in sorted(need):
        q = owner_of(j, n_total, world)
        if q != rank:
            recv.setdefault(q, []).append(j)
    for q in range(world):
        if q == rank:
            continue
        qf, qc = block_partition(n_total, world, q)
        wanted = set()
        for k in range(qf, qf + qc):
            for j in neighbours_fn(k, n_total, n_nbr):
                if first <= j < first + count:
                    wanted.add(j)
        if wanted:
            send[q] = sorted(wanted)
    boundary = sorted({j for lst in send.values() for j in lst})
    bset = set(boundary)
    # what every rank contributes to the boundary all-gather (the same on all ranks): its own keyframes that some other
    # rank's K4 reads, ascending; `contrib_count` = the common (padded) length
    contrib = {}
    for q in range(world):
        qf, qc = block_partition(n_total, world, q)
        wanted = set()
        for k in range(n_total):
            if qf <= k < qf + qc:
                continue
            for j in neighbours_fn(k, n_total, n_nbr):
                if qf <= j < qf + qc:
                    wanted.add(j)
        contrib[q] = sorted(wanted)
    assert contrib[rank] == boundary
    contrib_count = max([len(v) for v in contrib.values()] + [1])
    interior = [k for k in own if k not in bset]
    inputs = sorted(need)
    slot = {k: i for i, k in enumerate(inputs)}
    # K4 of a keyframe whose neighbours are all this rank's own needs nothing from the exchange: it runs BEFORE the wait
    # (`check_early`), only the others after it (`check_late`) -- with index-local covisibility these are the interior
    # and the boundary keyframes again
    oset = set(own)
    late = [k for k, row in zip(own, nbrs) if any(j not in oset for j in row)]
    lset = set(late)
    early = [k for k in own if k not in lset]
    return dict(first=first, count=count, own=own, nbrs=nbrs, inputs=inputs, boundary=boundary,
                interior=interior, recv=recv, send=send, n_total=n_total, world=world, rank=rank,
                slot=slot, n_slots=len(inputs), first_slot=slot[first], contrib=contrib, contrib_count=contrib_count,
                own_slots=[slot[k] for k in own], nbr_slots=[[slot[j] for j in row] for row in nbrs],
                boundary_slots=[slot[k] for k in boundary], interior_slots=[slot[k] for k in interior],
                check_early=early, check_late=late, check_early_slots=[slot[k] for k in early],
                check_late_slots=[slot[k] for k in late])


def _runs(idx):
    """sorted index list -> list of (start, stop) contiguous runs"""
    runs = []
    for i in idx:
        if runs and runs[-1][1] == i:
            runs[-1][1] = i + 1
        else:
            runs.append([i, i + 1])
    return [tuple(r) for r in runs]


def _staged(pool, group):
    """True when the pool lives on a GPU but the process group is gloo, which has no device
    point-to-point: the exchange is then staged through host memory.  This exists to rehearse the
    multi-rank flow with several ranks on ONE GPU (tests/test_gpu_shard.py, bench.py with
    SDM_BENCH_REHEARSE=1); production runs use RCCL and never take this path."""
    return pool.is_cuda and dist.get_backend(group) == "gloo"


class _StagedRecv:
    def __init__(self, work, buf, dst):
        self.work, self.buf, self.dst = work, buf, dst

    def wait(self):
        self.work.wait()
        self.dst.copy_(self.buf)


def exchange_halo_async(pool, pl, group=None):
    """torch transport: starts the point-to-point exchange of the boundary maps; returns a list of work
    handles (empty if there is nothing to exchange).  `pool` is indexed by LOCAL slot; contiguous runs
    of keyframes are sent/received as views of it (zero copy).  Call wait_all() before K4."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return []
    staged = _staged(pool, group)
    slot = pl["slot"]
    ops, recvs = [], []
    for peer in sorted(set(pl["send"]) | set(pl["recv"])):
        for a, b in _runs([slot[k] for k in pl["send"].get(peer, [])]):
            src = pool[a:b].cpu() if staged else pool[a:b]  # .cpu() waits for the producing kernels
            ops.append(dist.P2POp(dist.isend, src, peer, group=group))
        for a, b in _runs([slot[k] for k in pl["recv"].get(peer, [])]):
            dst = torch.empty(pool[a:b].shape, dtype=pool.dtype) if staged else pool[a:b]
            recvs.append((len(ops), dst, pool[a:b]))
            ops.append(dist.P2POp(dist.irecv, dst, peer, group=group))
    if not ops:
        return []
    works = dist.batch_isend_irecv(ops)
    if staged:  # gloo returns one handle per op
        assert len(works) == len(ops)
        for i, buf, dst in recvs:
            works[i] = _StagedRecv(works[i], buf, dst)
    return works


def wait_all(works):
    for w in works:
        w.wait()


def allgather_depth(pool, pl, group=None, gather=None):
    """torch transport: all-gather of every rank's block into `gather` ([n_total, H, W, 2], allocated on
    first use and returned for reuse), then the maps this rank's K4 reads are copied to their local slots."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return gather
    staged = _staged(pool, group)
    fs, cnt = pl["first_slot"], pl["count"]
    shape = (pl["n_total"],) + tuple(pool.shape[1:])
    if gather is None or tuple(gather.shape) != shape:
        gather = torch.empty(shape, dtype=pool.dtype, device="cpu" if staged else pool.device)
    mine = pool[fs:fs + cnt]
    dist.all_gather_into_tensor(gather, mine.cpu() if staged else mine, group=group)
    own = set(pl["own"])
    for a, b in _runs([k for k in pl["inputs"] if k not in own]):
        pool[pl["slot"][a]:pl["slot"][a] + (b - a)].copy_(gather[a:b])
    return gather


def halo_lists(pl):
    """(send, recv) as lists of (peer, local slot) for the engine's native exchange"""
    slot = pl["slot"]
    send = [(p, slot[k]) for p in sorted(pl["send"]) for k in pl["send"][p]]
    recv = [(p, slot[k]) for p in sorted(pl["recv"]) for k in pl["recv"][p]]
    return send, recv


def fetch_list(pl):
    """whole-block all-gather: [(index in the gathered sequence = global keyframe, local slot)] of the maps K4 reads
    from other ranks"""
    own = set(pl["own"])
    return [(k, pl["slot"][k]) for k in pl["inputs"] if k not in own]


def contrib_slots(pl):
    """boundary all-gather: this rank's contribution as local slots, padded to the common count by repeating the last
    one (a rank at the end of the sequence has half as many boundary keyframes; a rank with none repeats its first slot)"""
    s = [pl["slot"][k] for k in pl["contrib"][pl["rank"]]] or [pl["first_slot"]]
    return s + [s[-1]] * (pl["contrib_count"] - len(s))


def contrib_fetch_list(pl):
    """boundary all-gather: [(owner * contrib_count + position in the owner's contribution, local slot)] of the maps K4
    reads from other ranks"""
    out = []
    for q in sorted(pl["recv"]):
        pos = {k: i for i, k in enumerate(pl["contrib"][q])}
        out += [(q * pl["contrib_count"] + pos[k], pl["slot"][k]) for k in pl["recv"][q]]
    return out


def allgather_boundary(pool, pl, group=None):
    """torch transport of the boundary all-gather (rehearsal / fallback): pack, all_gather_into_tensor, unpack"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    staged = _staged(pool, group)
    idx = torch.tensor(contrib_slots(pl), dtype=torch.long, device=pool.device)
    mine = pool.index_select(0, idx)
    shape = (pl["world"] * pl["contrib_count"],) + tuple(pool.shape[1:])
    gather = torch.empty(shape, dtype=pool.dtype, device="cpu" if staged else pool.device)
    dist.all_gather_into_tensor(gather, mine.cpu() if staged else mine, group=group)
    for i, s in contrib_fetch_list(pl):
        pool[s].copy_(gather[i])


# ---- the compact wire format on host arrays ------------------------------------------------------------------------------
# What crosses ranks per map with sdm_exchange_compact(entries): the {rho,sigma} of the keyframe's active-list entries in
# list order, then a header of XCHG_HEADER float2 -- the list length and the 64-bit hash of the list as bit patterns
# (csrc/sdm_comm.h k_pack_lists / k_unpack_lists, csrc/sdm_ingest.h seg_hash_term).  The numpy statement below is the
# format's second implementation: the CPU tests move it between processes over gloo, the GPU tests check it against the
# engine's own packing byte for byte.
XCHG_HEADER = 8


def list_hash(lst):
    """hash of the pixel set the list holds: over the 64-pixel row segments it touches, the sum (mod 2^64) of the SplitMix64
    finalisation of (the segment's 64-bit membership mask) xor ((y << 16 | x0) * 0x9E3779B97F4A7C15), x0 = the segment's
    first column (csrc/sdm_ingest.h seg_hash_term)"""
    import numpy as np
    lst = np.asarray(lst, np.uint32).ravel()
    if lst.size == 0:
        return 0
    keys, inv = np.unique(lst & ~np.uint32(63), return_inverse=True)
    masks = np.zeros(keys.size, np.uint64)
    np.bitwise_or.at(masks, inv.ravel(), np.uint64(1) << (lst & np.uint32(63)).astype(np.uint64))
    with np.errstate(over="ignore"):
        z = masks ^ (keys.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return int(z.sum(dtype=np.uint64))


def pack_compact(depth_map, lst, entries):
    """depth_map: float32 [H, W, 2]; lst: uint32 (y << 16 | x); returns float32 [entries + XCHG_HEADER, 2]"""
    import numpy as np
    lst = np.asarray(lst, np.uint32)
    if lst.size > entries:
        raise ValueError("list of %d entries does not fit the wire format (%d)" % (lst.size, entries))
    out = np.zeros((entries + XCHG_HEADER, 2), np.float32)
    out[:lst.size] = np.asarray(depth_map, np.float32)[lst >> 16, lst & 0xFFFF]
    h = list_hash(lst)
    out.view(np.uint32)[entries] = (lst.size, h & 0xFFFFFFFF)
    out.view(np.uint32)[entries + 1] = (h >> 32, 0)
    return out


def unpack_compact(payload, lst, entries, depth_map):
    """scatters a payload through the receiver's list into depth_map (float32 [H, W, 2], zero outside the list);
    returns False -- and leaves the map alone -- when the payload was packed with another list"""
    import numpy as np
    lst = np.asarray(lst, np.uint32)
    u = np.asarray(payload, np.float32).view(np.uint32)
    h = list_hash(lst)
    if (int(u[entries, 0]), int(u[entries, 1]), int(u[entries + 1, 0])) != (lst.size, h & 0xFFFFFFFF, h >> 32):
        return False
    depth_map[lst >> 16, lst & 0xFFFF] = np.asarray(payload, np.float32)[:lst.size]
    return True


class HostCompactCodec:
    """pack / deliver on host arrays (CPU rehearsal): pool is a CPU tensor [slots, H, W, 2], lists[slot] the slot's list"""

    def __init__(self, pool, lists, entries):
        self.pool, self.lists, self.entries, self.refused = pool, lists, entries, 0

    def pack(self, slot):
        return torch.from_numpy(pack_compact(self.pool[slot].numpy(), self.lists[slot], self.entries))

    def deliver(self, slot, payload):
        if not unpack_compact(payload.numpy(), self.lists[slot], self.entries, self.pool[slot].numpy()):
            self.refused += 1


class EngineCompactCodec:
    """pack / deliver through the engine's own kernels and host memory (sdm_compact_pack_host / _unpack_host)"""

    def __init__(self, eng):
        self.eng, self.entries, self.refused = eng, eng.compact_entries, 0

    def pack(self, slot):
        return torch.from_numpy(self.eng.compact_pack_host(slot))

    def deliver(self, slot, payload):
        if self.eng.compact_unpack_host(slot, payload.numpy()):
            self.refused += 1


def exchange_halo_compact(codec, pl, group=None):
    """torch transport, compact payloads staged through host memory: the point-to-point form"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    slot = pl["slot"]
    ops, recvs = [], []
    for peer in sorted(set(pl["send"]) | set(pl["recv"])):
        for k in pl["send"].get(peer, []):
            ops.append(dist.P2POp(dist.isend, codec.pack(slot[k]), peer, group=group))
        for k in pl["recv"].get(peer, []):
            buf = torch.empty((codec.entries + XCHG_HEADER, 2), dtype=torch.float32)
            recvs.append((slot[k], buf))
            ops.append(dist.P2POp(dist.irecv, buf, peer, group=group))
    if ops:
        wait_all(dist.batch_isend_irecv(ops))
    for s, buf in recvs:
        codec.deliver(s, buf)


def allgather_boundary_compact(codec, pl, group=None):
    """torch transport, compact payloads staged through host memory: the boundary all-gather"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    # real contributions are packed; padding positions (a rank with fewer boundary keyframes than the common count, or none)
    # carry a zero payload -- nobody fetches them, and packing a real slot there would fail on a rank whose padding slot is not
    # reconstructed yet while its peers already sit in the all-gather
    real = [pl["slot"][k] for k in pl["contrib"][pl["rank"]]]
    packed = [codec.pack(s) for s in real]
    zero = torch.zeros((codec.entries + XCHG_HEADER, 2), dtype=torch.float32)
    mine = torch.stack(packed + [zero] * (pl["contrib_count"] - len(packed)))
    gather = torch.empty((pl["world"] * pl["contrib_count"],) + tuple(mine.shape[1:]), dtype=torch.float32)
    dist.all_gather_into_tensor(gather, mine, group=group)
    for i, s in contrib_fetch_list(pl):
        codec.deliver(s, gather[i])


def setup_native_comm(eng, group=None):
    """Builds the engine's RCCL communicator: rank 0 draws the unique id and the bytes travel over the
    existing torch.distributed group (any channel would do; the C++ drop-in would use a file or MPI)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [eng.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    eng.comm_init(box[0], world, rank)


def agree_compact_wire(eng, pl, group=None, max_fraction=0.5):
    """Chooses the wire format of the native exchange for this job (every rank calls this; one all-reduce): the maps that
    cross ranks travel as the {rho,sigma} of their keyframe's active-list entries (sdm_exchange_compact) if the longest
    list among ALL ranks' keyframes, rounded up to 64 entries, is at most `max_fraction` of the pixels -- else whole maps.
    Returns the entries per map (0 = whole maps)."""
    longest = max([eng.active_count(s) for s in range(pl["n_slots"])] + [0])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor([longest], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        longest = int(t.item())
    entries = (longest + 63) // 64 * 64
    if entries == 0 or entries > max_fraction * eng.W * eng.H:
        entries = 0
    eng.exchange_compact(entries)
    return entries


_gather_cache = {}


def sub_blocks(count, pieces):
    """(offset, count) of `pieces` nearly equal contiguous sub-blocks of a block of `count` keyframes"""
    pieces = max(1, min(pieces, count))
    base, extra = divmod(count, pieces)
    out, off = [], 0
    for i in range(pieces):
        c = base + (1 if i < extra else 0)
        out.append((off, c))
        off += c
    return out


AG_PIECES = 4  # sub-blocks of the pipelined all-gather (native transport)


def pipeline_step(eng, pool, pl, min_d, max_d, exchange="allgather", group=None, transport="torch", ag_pieces=AG_PIECES,
                  force_pieces=False):
    """One pass of the hot path over this rank's keyframe block (what bench.py times and the
    multi-rank tests check): SemiDenseRecon (K1-K3) -> exchange of {rho,sigma} maps -> inter-keyframe
    check (K4, snapshot form) + point set (K5; back-projected in the checking kernel).

    exchange: "allgather" (boundary keyframes, overlapped with the interior keyframes' K1-K3 and K4), "allgather_late" (the
    same collective after an unsplit reconstruction, overlapped with the local K4 only), "allgather_full" (whole block,
    pipelined in sub-blocks) or "halo" (point-to-point); see the module docstring.  force_pieces: run the all-gather bookkeeping at world size 1
    (the tests' rehearsal)."""
    world = pl["world"]
    own, nbrs = pl["own_slots"], pl["nbr_slots"]
    nb_of = dict(zip(own, nbrs))
    boundary, interior = pl["boundary_slots"], pl["interior_slots"]
    early, late = pl["check_early_slots"], pl["check_late_slots"]
    native = transport == "native"
    rot_of = pl.get("rot_of")  # optional {slot: [median in-plane rotation per neighbour, degrees]} (PM.cc:170-179)

    def recon(slots, nb, mn, mx):
        eng.recon(slots, nb, mn, mx, rot=None if rot_of is None else [rot_of[k] for k in slots])

    def check(slots):  # K4 with K5 riding along (PM.cc:300-306); snapshot form: the pool's maps stay as reconstructed
        if slots:
            eng.inter_check_pointset(slots, [nb_of[k] for k in slots], commit=False)

    # In every sharded form the keyframes whose K4 reads no other rank's map are checked BEFORE the wait for the exchange
    # (the transfer has the interior keyframes' K1-K3 and their K4 to hide behind), the others after it.
    if world > 1 and exchange == "halo" and boundary:
        recon(boundary, [nb_of[k] for k in boundary], min_d, max_d)
        compact = (not native) and getattr(eng, "compact_entries", 0) > 0
        if native:
            eng.exchange_halo_begin(*halo_lists(pl))
        elif compact:
            works = None
        else:
            works = exchange_halo_async(pool, pl, group)
        if interior:
            recon(interior, [nb_of[k] for k in interior], min_d, max_d)
        check(early)
        if native:
            eng.exchange_wait()
        elif compact:  # host-staged compact payloads (blocking: a rehearsal transport)
            codec = EngineCompactCodec(eng)
            exchange_halo_compact(codec, pl, group)
            eng.staged_refused = getattr(eng, "staged_refused", 0) + codec.refused
        else:
            wait_all(works)
            eng.mark_depth_present([s for _, s in halo_lists(pl)[1]])
        check(late)
    elif exchange == "allgather" and (world > 1 or force_pieces):
        # all-gather of the maps that cross ranks: reconstruct the boundary keyframes, start the collective on the
        # engine's exchange stream, reconstruct the interior keyframes meanwhile
        if boundary:
            recon(boundary, [nb_of[k] for k in boundary], min_d, max_d)
        else:
            recon(own[:1], nbrs[:1], min_d, max_d)  # nothing crosses ranks: the (padded) contribution still needs a map
        rest = interior if boundary else own[1:]
        if native:
            eng.allgather_begin(pl["contrib_count"])
            eng.allgather_piece(contrib_slots(pl))
            if rest:
                recon(rest, [nb_of[k] for k in rest], min_d, max_d)
            check(early)
            eng.allgather_finish(contrib_fetch_list(pl))
        elif getattr(eng, "compact_entries", 0) > 0:
            codec = EngineCompactCodec(eng)
            allgather_boundary_compact(codec, pl, group)
            eng.staged_refused = getattr(eng, "staged_refused", 0) + codec.refused
            if rest:
                recon(rest, [nb_of[k] for k in rest], min_d, max_d)
            check(early)
        else:
            allgather_boundary(pool, pl, group)
            if rest:
                recon(rest, [nb_of[k] for k in rest], min_d, max_d)
            check(early)
            eng.mark_depth_present([s for _, s in contrib_fetch_list(pl)])
        check(late)
    elif exchange == "allgather_late" and (world > 1 or force_pieces):
        # the boundary all-gather WITHOUT splitting the reconstruction: K1-K3 over the whole block in one set of launches (no
        # second K1 launch with its tail), then the collective with only the local keyframes' K4 to hide behind.  Cheaper on
        # the compute side, a third of the window: the better schedule when the transfer is short (compact wire format).
        recon(own, nbrs, min_d, max_d)
        if native:
            eng.allgather_begin(pl["contrib_count"])
            eng.allgather_piece(contrib_slots(pl))
            check(early)
            eng.allgather_finish(contrib_fetch_list(pl))
        elif getattr(eng, "compact_entries", 0) > 0:
            codec = EngineCompactCodec(eng)
            allgather_boundary_compact(codec, pl, group)
            eng.staged_refused = getattr(eng, "staged_refused", 0) + codec.refused
            check(early)
        else:
            allgather_boundary(pool, pl, group)
            check(early)
            eng.mark_depth_present([s for _, s in contrib_fetch_list(pl)])
        check(late)
    elif exchange == "allgather_full" and native and ag_pieces > 1 and (world > 1 or force_pieces):
        # whole-block all-gather, pipelined: the block is reconstructed in sub-blocks; each one's maps are gathered on
        # the engine's exchange stream while the next one's K1-K3 run (sdm_allgather_begin / _piece / _finish)
        eng.allgather_begin(pl["count"])
        for off, cnt in sub_blocks(pl["count"], ag_pieces):
            recon(own[off:off + cnt], nbrs[off:off + cnt], min_d, max_d)
            eng.allgather_piece(own[off:off + cnt])
        check(early)
        # position in the owner's block == keyframe - owner's first keyframe; owner * count + position == keyframe
        eng.allgather_finish(fetch_list(pl))
        check(late)
    else:
        recon(own, nbrs, min_d, max_d)
        if world > 1:
            if native:
                eng.allgather_depth(pl["first_slot"], pl["count"], fetch_list(pl))
            else:
                key = (id(pool), pl["n_total"])
                _gather_cache[key] = allgather_depth(pool, pl, group, _gather_cache.get(key))
                eng.mark_depth_present([s for _, s in fetch_list(pl)])
        eng.inter_check_pointset(own, nbrs, commit=False)
