"""Keyframe sharding for the multi-GPU path.

K1-K3 (SemiDenseRecon, PM.cc:137-256) of keyframe k read only immutable inputs of k and its
covisible neighbours; K4 (InterKeyFrameDepthChecking, PM.cc:628-799) needs the neighbours'
FINISHED {rho, sigma} maps (the reference gates on that at PM.cc:292-298).  So keyframes shard as
contiguous blocks, one block per GPU, with exactly one exchange step between K3 and K4.

Slot numbering is GLOBAL (slot == keyframe index) on every rank and the depth pool is one torch
tensor [n_total, H, W, 2] handed to the engine as `ext_depth_pool`, so exchanged maps land where K4
reads them and no re-indexing or staging copy is needed.  Two exchange forms:

  halo (default)  each rank receives only the maps its K4 will read (the covisible neighbours that
                  live on other ranks: N/2 keyframes from each adjacent block for an index-local
                  covisibility graph) with batched point-to-point send/recv.  xGMI is point-to-point,
                  so this moves 2 x (N/2) x 8P bytes per rank over two direct links, independent of
                  the number of GPUs, and it is issued right after the boundary keyframes are
                  reconstructed so it overlaps the reconstruction of the interior ones.
  allgather       the whole pool, in place (BASELINE.json's wording); (world-1) x block bytes per rank.
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
    """Returns dict(first, count, own, nbrs, inputs, boundary, interior, recv, send).

    inputs   slots whose IMAGES this rank must hold = own block + its neighbours (input halo)
    boundary own keyframes some OTHER rank's K4 reads (reconstruct these first, then exchange)
    interior the rest of the own block
    recv     {peer: sorted keyframes owned by peer that this rank's K4 reads}
    send     {peer: sorted own keyframes that peer's K4 reads}
    Every rank derives every other rank's needs from the same deterministic neighbour function, so
    send/recv lists match pairwise without negotiation."""
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
    interior = [k for k in own if k not in bset]
    return dict(first=first, count=count, own=own, nbrs=nbrs, inputs=sorted(need), boundary=boundary,
                interior=interior, recv=recv, send=send)


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
    SDM_BENCH_REHEARSE=1); production runs use RCCL ("nccl") and never take this path."""
    return pool.is_cuda and dist.get_backend(group) == "gloo"


class _StagedRecv:
    def __init__(self, work, buf, dst):
        self.work, self.buf, self.dst = work, buf, dst

    def wait(self):
        self.work.wait()
        self.dst.copy_(self.buf)


def exchange_halo_async(pool, pl, group=None):
    """Starts the point-to-point exchange of the boundary maps; returns a list of work handles
    (empty if there is nothing to exchange).  Contiguous runs of keyframes are sent/received as
    views of `pool` (zero copy).  Call wait_all() before K4."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return []
    staged = _staged(pool, group)
    ops, recvs = [], []
    for peer in sorted(set(pl["send"]) | set(pl["recv"])):
        for a, b in _runs(pl["send"].get(peer, [])):
            src = pool[a:b].cpu() if staged else pool[a:b]  # .cpu() waits for the producing kernels
            ops.append(dist.P2POp(dist.isend, src, peer, group=group))
        for a, b in _runs(pl["recv"].get(peer, [])):
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


def allgather_depth(pool, first, count, group=None):
    """In-place all-gather of the depth pool: this rank has just written rows [first, first+count).
    After the call every rank holds every keyframe's {rho, sigma}."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    if _staged(pool, group):
        host = torch.empty(pool.shape, dtype=pool.dtype)
        dist.all_gather_into_tensor(host, pool[first:first + count].cpu(), group=group)
        pool.copy_(host)
        return
    mine = pool[first:first + count]
    dist.all_gather_into_tensor(pool, mine, group=group)


def pipeline_step(eng, pool, pl, min_d, max_d, exchange="halo", group=None):
    """One pass of the hot path over this rank's keyframe block (what bench.py times and the
    multi-rank tests check): SemiDenseRecon (K1-K3) -> exchange of {rho,sigma} maps -> inter-keyframe
    check (K4, snapshot form) + point set (K5; back-projected in the checking kernel).

    halo: boundary keyframes are reconstructed first; their maps travel to the adjacent ranks
    (point-to-point over xGMI) while the interior keyframes are reconstructed."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    own, nbrs = pl["own"], pl["nbrs"]
    nb_of = dict(zip(own, nbrs))
    boundary, interior = pl["boundary"], pl["interior"]
    if world > 1 and exchange == "halo" and boundary:
        eng.recon(boundary, [nb_of[k] for k in boundary], min_d, max_d)
        works = exchange_halo_async(pool, pl, group)
        if interior:
            eng.recon(interior, [nb_of[k] for k in interior], min_d, max_d)
        wait_all(works)
    else:
        eng.recon(own, nbrs, min_d, max_d)
        if world > 1:
            allgather_depth(pool, pl["first"], pl["count"], group)
    eng.inter_check_pointset(own, nbrs, commit=False)  # K4 with K5 riding along (PM.cc:300-306)
